cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do
  for cg in 0 1; do
    SMCP_CONTIG=$cg timeout -k 10 200 python3 bench.py --no-secondary --steps 10 --warmup 3 --no-cpu --no-back-solve --tune-placement 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('contig $cg:', d['value'], d['ms_per_step'], d['kernel_ms_per_step']['k_fam_terms'], d['kernel_ms_per_step']['k_lf_assemble_lds_dyn'])"
  done
done

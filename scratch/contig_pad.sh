# contiguous physical memory (the reproducible worst case for k_fam_terms): does padding the per-right-hand-side stride of the
# exchange buffer help?
cd $GRAFT_REPO_ROOT
for pad in 0 32 512 2048 8192 65536 262144 1048576 48 1584; do
  SMCP_CONTIG=1 SMCP_UPDP_PAD=$pad timeout -k 10 200 python3 bench.py --no-secondary --steps 6 --warmup 2 --no-cpu --no-back-solve --tune-placement 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('contig pad $pad:', d['ms_per_step'], d['kernel_ms_per_step']['k_fam_terms'], d['kernel_ms_per_step']['k_lf_assemble_lds_dyn'])"
done

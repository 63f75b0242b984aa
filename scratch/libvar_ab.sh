# A/B of library variants (scratch/libvar/*.so), three bench runs each: bash scratch/libvar_ab.sh
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/libvar_ab.txt
for rep in 1 2 3; do
for V in base aux1 aux16 aux17; do
  cp scratch/libvar/$V.so smcp_amd/libsmcp_amd.so
  timeout -k 10 200 python3 bench.py --no-cpu --no-secondary --steps 10 > gpurun_out/fp.json 2>/dev/null || exit 1
  python3 -c "
import json
d=json.loads(open('gpurun_out/fp.json').read().strip().splitlines()[-1])
k=d['kernel_ms_per_step']
print('$V','ms/step',d['ms_per_step'],'terms',k.get('k_fam_terms'),'asm',k.get('k_lf_assemble_lds'),'gram',k.get('k_gram_diag128'))
" >> gpurun_out/libvar_ab.txt
done; done
sort gpurun_out/libvar_ab.txt

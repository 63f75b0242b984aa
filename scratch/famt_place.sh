# placement study of k_fam_terms: the variants scratch/libvar/pad_N.so (N s_nop words ahead of the kernel body, kernel aligned to 4 KB)
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/famt_place.txt
for P in 0 64 128 192; do
  cp scratch/libvar/pad_$P.so smcp_amd/libsmcp_amd.so
  timeout -k 10 200 python3 bench.py --no-cpu --no-secondary --steps 10 > gpurun_out/fp.json 2>/dev/null || exit 1
  python3 -c "
import json
d=json.loads(open('gpurun_out/fp.json').read().strip().splitlines()[-1])
k=d['kernel_ms_per_step']
print('pad',$P,'ms/step',d['ms_per_step'],'terms',k.get('k_fam_terms'),'prep',k.get('k_famt_prep'))
" >> gpurun_out/famt_place.txt
done
cat gpurun_out/famt_place.txt

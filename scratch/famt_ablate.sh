# ablations of k_fam_terms (SMCP_SKIP bits: 2 = no stores, 8 = no products): bash scratch/famt_ablate.sh
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/famt_ablate.txt
for sk in 0 2 8 10; do
  SMCP_SKIP=$sk timeout -k 10 200 python3 bench.py --no-cpu --no-secondary --steps 10 > gpurun_out/fa_$sk.json 2>/dev/null || exit 1
  python3 -c "
import json
d=json.loads(open('gpurun_out/fa_$sk.json').read().strip().splitlines()[-1])
k=d['kernel_ms_per_step']
print('skip',$sk,'ms/step',d['ms_per_step'],'terms',k.get('k_fam_terms'),'prep',k.get('k_famt_prep'))
" >> gpurun_out/famt_ablate.txt
done
cat gpurun_out/famt_ablate.txt

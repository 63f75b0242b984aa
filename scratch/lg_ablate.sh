# ablations of k_leaf_pairs (SMCP_LGSKIP bits): bash scratch/lg_ablate.sh -> gpurun_out/lg_ablate.txt
cd $GRAFT_REPO_ROOT
for sk in 0 1 2 3 4; do
  SMCP_LGSKIP=$sk timeout -k 10 200 python3 bench.py --no-cpu --steps 10 > gpurun_out/lg_$sk.json 2>/dev/null || exit 1
  python3 -c "
import json
d=json.loads(open('gpurun_out/lg_$sk.json').read().strip().splitlines()[-1])
k=d['kernel_ms_per_step']
print('skip',$sk,'ms/step',d['ms_per_step'],'pairs',k.get('k_leaf_pairs'),'tables',k.get('k_leaf_tables'),'reduce',k.get('k_gram_reduce'))
" >> gpurun_out/lg_ablate.txt
done
cat gpurun_out/lg_ablate.txt

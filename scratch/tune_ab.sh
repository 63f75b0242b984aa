# placement tuning with both buffers: bash scratch/tune_ab.sh
for i in 1 2 3 4 5 6; do
  timeout -k 10 120 python bench.py --no-secondary --no-cpu --no-back-solve --steps 8 --warmup 2 > gpurun_out/tune_ab$i.json 2>/dev/null || exit 1
  python3 - <<PY
import json
d=json.loads(open('gpurun_out/tune_ab$i.json').read().strip().splitlines()[-1])
print('run $i', d['value'], d['ms_per_step'], d['kernel_ms_per_step'].get('k_fam_terms'), d['config'].get('placement_tuning'))
PY
done

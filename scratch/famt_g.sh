# k_fam_terms: right-hand-side slices per family (grid y) -- bash scratch/famt_g.sh
for g in 0 1 2 3 4; do
  SMCP_FAMT_G=$g timeout -k 10 120 python bench.py --no-secondary --no-cpu --no-back-solve --steps 8 --warmup 2 > gpurun_out/famt_g$g.json 2>/dev/null || exit 1
  python3 - <<PY
import json
d=json.loads(open('gpurun_out/famt_g$g.json').read().strip().splitlines()[-1])
print('g=$g', d['value'], d['ms_per_step'], d['kernel_ms_per_step'].get('k_fam_terms'), d['config'].get('placement_tuning'))
PY
done

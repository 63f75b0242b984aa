# sixteen-wave tile products for launches of up to N workgroups per CU: bash scratch/pd_wgs.sh
for rep in 1 2; do
for w in 1 4 8 16; do
  SMCP_PD_WGS=$w timeout -k 10 120 python bench.py --no-secondary --no-cpu --steps 8 --warmup 2 > gpurun_out/pd_wgs$w.json 2>/dev/null || exit 1
  python3 - <<PY
import json
d=json.loads(open('gpurun_out/pd_wgs$w.json').read().strip().splitlines()[-1])
k=d['kernel_ms_per_step']
print('wgs=$w', d['value'], d['ms_per_step'], d['back_solve']['ms'], {x:k.get(x) for x in ['k_fam_terms','k_lf_up1','k_lf_up2','k_lf_up3','k_lf_down1','k_lf_down3']}, d['config']['placement_tuning']['probe_ms_after'])
PY
done
done

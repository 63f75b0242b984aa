# cache policy of k_fam_terms' result stores: libs built beforehand as smcp_amd/libsmcp_amd.so.auxN -- bash scratch/famt_aux.sh
cp smcp_amd/libsmcp_amd.so /tmp/lib_orig.so
for rep in 1 2; do
for aux in 0 2; do
  cp smcp_amd/libsmcp_amd.so.aux$aux smcp_amd/libsmcp_amd.so
  timeout -k 10 120 python bench.py --no-secondary --no-cpu --no-back-solve --steps 8 --warmup 2 > gpurun_out/famt_aux$aux.json 2>/dev/null || exit 1
  python3 - <<PY
import json
d=json.loads(open('gpurun_out/famt_aux$aux.json').read().strip().splitlines()[-1])
print('aux=$aux', d['value'], d['ms_per_step'], d['kernel_ms_per_step'].get('k_fam_terms'), d['kernel_ms_per_step'].get('k_lf_assemble_lds_dyn'), d['kernel_ms_per_step'].get('k_gram_diag128'), d['config'].get('placement_tuning'))
PY
done
done
cp /tmp/lib_orig.so smcp_amd/libsmcp_amd.so

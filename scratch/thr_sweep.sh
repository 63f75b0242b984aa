#!/bin/bash
# timing study: threads of the one-workgroup diagonal step (config 2) and of potrf(H) (synth50k)
set -e
mkdir -p gpurun_out
for t in 256 512 1024; do
  SMCP_DIAG_THREADS=$t python bench.py --workload dense4096 --no-cpu --steps 3 > gpurun_out/thr_diag_$t.json 2>/dev/null
  SMCP_POTRF_THREADS=$t python bench.py --no-cpu > gpurun_out/thr_potrf_$t.json 2>/dev/null
done
python - <<'PY'
import json
for t in (256, 512, 1024):
    d = json.load(open("gpurun_out/thr_diag_%d.json" % t)); k = d["kernel_ms_per_step"]
    print("diag threads %4d: dense4096 %.3f ms/step  k_lf_diag %.3f" % (t, d["ms_per_step"], k.get("k_lf_diag", 0)))
    d = json.load(open("gpurun_out/thr_potrf_%d.json" % t)); k = d["kernel_ms_per_step"]
    print("potrf threads %4d: synth50k %.3f ms/step  k_dense_potrf %.4f k_lf_diag %.4f" % (t, d["ms_per_step"], k.get("k_dense_potrf", 0), k.get("k_lf_diag", 0)))
PY

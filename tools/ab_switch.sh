#!/bin/bash
# A / B of one library switch on the GPU-only bench line of a workload: tools/ab_switch.sh SMCP_UP_FUSED "0 1" [workload] [steps]
# (each setting twice, interleaved; prints ms_per_step, the stage times and the kernels that moved)
set -o pipefail
SW=$1; VALS=${2:-"0 1"}; WL=${3:-synth50k}; STEPS=${4:-20}
mkdir -p gpurun_out/ab
for rep in $(seq 1 ${REPS:-2}); do
  for v in $VALS; do
    env $SW=$v python3 bench.py --workload $WL --steps $STEPS --warmup 3 --no-cpu --no-secondary > gpurun_out/ab/${SW}_${v}_${rep}.json 2> gpurun_out/ab/${SW}_${v}_${rep}.err || { echo "bench failed ($SW=$v)"; tail -5 gpurun_out/ab/${SW}_${v}_${rep}.err; exit 1; }
    python3 - gpurun_out/ab/${SW}_${v}_${rep}.json $SW $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d.get("kernel_ms_per_step") or {}
top = ", ".join("%s %.3f" % (n, ms) for n, ms in list(k.items())[:12])
print("%s=%s: %.4f ms/step  %.1f/s  stages %s  back_solve %s" % (sys.argv[2], sys.argv[3], d["ms_per_step"], d["value"],
      {a: b for a, b in (d.get("stages") or {}).items() if a != "what"}, (d.get("back_solve") or {}).get("ms")))
print("   ", top)
PY
  done
done

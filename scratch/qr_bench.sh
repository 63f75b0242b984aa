#!/bin/bash
# kktsolver=qr bench lines for the two tall triangular solves
for v in "SMCP_QR_TRSM=mfma" "SMCP_QR_TRSM=fma" "SMCP_QR_TRSM=mfma"; do
  env $v timeout -k 10 300 python bench.py --kktsolver qr --steps 10 --warmup 3 > gpurun_out/bq.json 2> gpurun_out/bq.err || exit 1
  python - "$v" <<PY
import json, sys
d=json.loads(open("gpurun_out/bq.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"], "trsm", d["kernel_ms_per_step"].get("k_stack_trsm"))
PY
done

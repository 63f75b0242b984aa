#!/bin/bash
# A/B of library builds: swaps smcp_amd/libsmcp_amd.so on the GPU box copy only
set -e
cp smcp_amd/libsmcp_amd.so /tmp/orig.so
for rep in 1 2; do
for v in orig nw12 nw16; do
  if [ $v = orig ]; then cp /tmp/orig.so smcp_amd/libsmcp_amd.so; else cp tools/micro/bin/libsmcp_$v.so smcp_amd/libsmcp_amd.so; fi
  python3 bench.py --workload synth50k --steps 20 --warmup 3 --no-cpu --no-secondary > /tmp/o.json 2>/tmp/o.err
  python3 - $v <<'PY'
import json, sys
d = json.loads(open('/tmp/o.json').read().strip().splitlines()[-1])
k = d["kernel_ms_per_step"]
print("%s: %.4f ms/step  k_fam_terms %.3f  k_lf_assemble_fz %.3f  famt_prep %.3f  leaf_pairs %.3f" % (sys.argv[1], d["ms_per_step"], k.get("k_fam_terms", 0), k.get("k_lf_assemble_fz", 0), k.get("k_famt_prep", 0), k.get("k_leaf_pairs", 0)))
PY
done
done
cp /tmp/orig.so smcp_amd/libsmcp_amd.so

#!/bin/bash
for st in 0 2 4 6 8 12; do
  SMCP_FAM2_STAG=$st python bench.py --steps 10 --warmup 3 --no-cpu 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('stagger $st', d['kernel_ms_per_step'].get('k_hess_up_fam'), d['value'])"
done

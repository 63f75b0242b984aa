#!/bin/bash
# timing ablations of k_fam_sparse (results are WRONG with any switch on): SMCP_SKIP bits 1 = no child panel stores,
# 2 = no parent stores, 4 = no children work, 8 = no MFMA phases
for sk in 0 4 8 12; do
  SMCP_SKIP=$sk python bench.py --steps 10 --warmup 3 --no-cpu 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('skip $sk', d['kernel_ms_per_step'].get('k_hess_up_fam'))"
done

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/gputests.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/gputests.log
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"; grep -i "unknown switch\|error" gpurun_out/bench_default.err | head
python3 -c "
import json
d=json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['value_tuned'], d['value_as_allocated'], d['ms_per_step'], d['back_solve']['ms'], d['cpu_baseline']['gpu_vs_oracle_relerr'], d['cpu_baseline'].get('schur_vs_oracle_relerr'))
r=d['roofline']; print({k:r[k] for k in r if k!='note'})
for n,s in d['secondary'].items(): print(n, s.get('value'), s.get('ms_per_step'), s.get('error'))
"
SMCP_TYPO=1 python3 -c "
import torch
from smcp_amd import problems
from smcp_amd.symbolic import Symbolic
s=Symbolic(problems.band_pattern(20,2)); s.device_init(0,1)
from smcp_amd import chordal
from smcp_amd.cspmatrix import cspmatrix
import numpy as np
X=cspmatrix(s, torch.from_numpy(problems.random_factor_blkval(s,0)).cuda()); chordal.llt(X); chordal.cholesky(X); print('typo check done')
" 2>&1 | tail -3

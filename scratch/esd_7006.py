"""Case 7006 of scratch/fuzz_ipm.py (embedding driver ends 'unknown' at iteration 16): the run with progress output."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from smcp_amd import base, solvers
from smcp_amd.symbolic import Symbolic
import fuzz_parity
case, seed0 = 6, 7000
rng = np.random.default_rng(seed0 + case)
kind = case % 4
pat = fuzz_parity.pattern(rng, [1, 3, 0][kind - 1])
nv = Symbolic(pat).nnz
m = int(min(rng.integers(2, 16), max(1, nv // 4)))
P = base.pattern_SDP(pat, m, density=float(rng.choice([0.01, 0.05, 0.2])), seed=int(rng.integers(1 << 30)))
solvers.options.update(show_progress=True, maxiters=150)
for k, v in (("default", {}), ("refinement 2", {"esd_kkt_refinement": 2})):
    solvers.options.update(v)
    print("====", k, flush=True)
    try:
        s = P.solve_esd(kktsolver="chol")
        print(s["status"], s["iterations"], s["primal objective"], s.get("gap"), s.get("primal infeasibility"), s.get("dual infeasibility"))
    except Exception as e:
        print("EXC", type(e).__name__, e)

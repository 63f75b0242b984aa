import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, scipy.sparse as sp, torch
from smcp_amd import base, solvers
import fuzz_parity
from smcp_amd.symbolic import Symbolic
case = int(sys.argv[1]) if len(sys.argv) > 1 else 34
rng = np.random.default_rng(9000 + case)
kind = case % 4
pat = fuzz_parity.pattern(rng, [1, 3, 0][kind - 1])
nv = Symbolic(pat).nnz
m = int(min(rng.integers(2, 16), max(1, nv // 4)))
dens = float(rng.choice([0.01, 0.05, 0.2])); seed = int(rng.integers(1 << 30))
def go():
    P = base.pattern_SDP(pat, m, density=dens, seed=seed)
    solvers.options.update(show_progress=True, maxiters=100)
    s = P.solve_esd()
    print(s["status"], s["iterations"])
if torch.cuda.is_available():
    go()
else:
    from oracle_backend import oracle_backend
    with oracle_backend():
        go()

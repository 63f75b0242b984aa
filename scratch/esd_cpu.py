import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from oracle_backend import oracle_backend
from smcp_amd import base, solvers
n, m, bw = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
scaling = sys.argv[4] if len(sys.argv) > 4 else "primal"
solvers.options.update(show_progress=True, maxiters=60, debug=(len(sys.argv) > 5))
with oracle_backend():
    P = base.band_SDP(n, m, bw, seed=0)
    sol = P.solve_esd(scaling=scaling)
    print(sol["status"], sol["iterations"], sol["dimacs"])

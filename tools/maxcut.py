import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from smcp_amd import base, solvers
from smcp_amd.symbolic import Symbolic
n, e = int(sys.argv[1]), int(sys.argv[2])
P = base.maxcut_SDP(n, e, seed=0)
solvers.options.update(show_progress=True, maxiters=int(os.environ.get("MAXIT", "60")))
t0 = time.time()
# strictly feasible dual start: S = C - Diag(y) > 0 with y = -(max row sum of |C|) - 1
C = P.get_A(0)
y0 = -np.ones(n) * (abs(C).sum(axis=1).max() + 1.0)
sol = P.solve_feas(scaling="dual", dualstart={"y": y0})
print("status", sol["status"], "iters", sol["iterations"], "pobj", sol["primal objective"], "dobj", sol["dual objective"],
      "time %.1fs" % (time.time() - t0), "dimacs", ["%.1e" % v for v in sol["dimacs"]])

import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from smcp_amd import base, solvers
solvers.options.update(show_progress=False)
P = base.band_SDP(200, 100, 3, seed=0)
X0 = P._X0
for sc in ("primal", "dual"):
    t0 = time.time()
    sol = P.solve_feas(scaling=sc, primalstart={"x": P._X0}, dualstart={"y": P._y0, "s": P._S0})
    dt = time.time() - t0
    print(sc, sol["status"], sol["iterations"], "%.2f s total, %.3f s/iteration" % (dt, dt / max(1, sol["iterations"])))

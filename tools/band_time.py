import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import scipy.sparse as sp
from smcp_amd import base, solvers
solvers.options.update(show_progress=False)
for am in (False, True):
    solvers.options["amalgamate"] = am
    for (n, m, bw) in ((200, 100, 3), (500, 100, 3)):
        P = base.band_SDP(n, m, bw, seed=0)
        for sc in ("primal", "dual"):
            t0 = time.time()
            sol = P.solve_feas(scaling=sc, primalstart={"x": P._X0}, dualstart={"y": P._y0, "s": P._S0})
            dt = time.time() - t0
            print("amalgamate", am, "band n=%d m=%d bw=%d feas %s:" % (n, m, bw, sc), sol["status"], sol["iterations"], "%.2f s total, %.4f s/iteration" % (dt, dt / max(1, sol["iterations"])), flush=True)
        t0 = time.time()
        sol = P.solve_esd()
        dt = time.time() - t0
        print("amalgamate", am, "band n=%d esd primal:" % n, sol["status"], sol["iterations"], "%.2f s total, %.4f s/iteration" % (dt, dt / max(1, sol["iterations"])), flush=True)

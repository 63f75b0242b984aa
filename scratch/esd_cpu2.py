import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from oracle_backend import oracle_backend
from smcp_amd import base, solvers
solvers.options.update(show_progress=False, maxiters=100)
with oracle_backend():
    for (n, m, bw) in ((60, 20, 3), (200, 100, 3), (100, 50, 5)):
        for sc in ("primal", "dual"):
            for kr in (0, 1, 2):
                for dsh in (True, False):
                    solvers.options.update(esd_kkt_refinement=kr, esd_ds_from_hessian=dsh)
                    P = base.band_SDP(n, m, bw, seed=0)
                    sol = P.solve_esd(scaling=sc)
                    print(n, m, bw, sc, "kktref", kr, "dsH", dsh, sol["status"], sol["iterations"], "gap %.1e pres %.1e dres %.1e" % (sol["gap"], sol["primal infeasibility"], sol["dual infeasibility"]), flush=True)

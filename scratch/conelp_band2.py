import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, scipy.sparse as sp
from smcp_amd import base, solvers
solvers.options.update(show_progress=False, maxiters=60, feastol=1e-6, abstol=1e-5, reltol=1e-5)
P = base.band_SDP(200, 100, 3, seed=0)
n, m = P.n, P.m
G = sp.hstack([sp.csc_matrix(P.get_A(i + 1).reshape((n * n, 1), order="F")) for i in range(m)]).tocsc()
h = np.asarray(P.get_A(0).todense()).reshape(-1, order="F")
for v in (True, False):
    for ft in (1e-6, 1e-7, 1e-8):
        solvers.options.update(esd_ds_from_hessian=v, feastol=ft, abstol=ft*10, reltol=ft*10)
        sol = solvers.conelp(-P.b, G, h, {"l": 0, "q": [], "s": [n]})
        print("ds_from_hessian", v, "feastol %.0e" % ft, sol["status"], sol["iterations"], "gap %.1e" % sol["gap"], "pres %.1e dres %.1e" % (sol["primal infeasibility"], sol["dual infeasibility"]))

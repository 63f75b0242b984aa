import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, scipy.sparse as sp
from smcp_amd import base, solvers
solvers.options.update(show_progress=True, maxiters=24, feastol=1e-8, abstol=1e-7, reltol=1e-7, debug=True)
P = base.band_SDP(200, 100, 3, seed=0)
n, m = P.n, P.m
G = sp.hstack([sp.csc_matrix(P.get_A(i + 1).reshape((n * n, 1), order="F")) for i in range(m)]).tocsc()
h = np.asarray(P.get_A(0).todense()).reshape(-1, order="F")
sol = solvers.conelp(-P.b, G, h, {"l": 0, "q": [], "s": [n]})

import sys, os; sys.path.insert(0, os.getcwd())
import numpy as np
from smcp_amd import solvers
rng = np.random.default_rng(0)
nvar, ncon = 6, 15
G = rng.standard_normal((ncon, nvar)); x0 = rng.standard_normal(nvar)
h = G @ x0 + rng.random(ncon) + 0.1; z0 = rng.random(ncon) + 0.1; c = -G.T @ z0
pass
sol = solvers.lp(c, G, h)
print(sol['status'])

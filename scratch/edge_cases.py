import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, scipy.sparse as sp
from smcp_amd import base, solvers
solvers.options.update(show_progress=False)
def run(name, P, **kw):
    try:
        sol = P.solve_feas(**kw) if kw.pop("feas", True) else P.solve_esd(**kw)
        print(name, sol["status"], sol["iterations"], "pobj %.6g dobj %.6g" % (sol["primal objective"], sol["dual objective"]))
    except Exception as e:
        print(name, "EXC", type(e).__name__, e)
for (n, m, bw) in ((10, 1, 2), (3, 1, 1), (2, 1, 0), (6, 3, 5), (40, 5, 0)):
    P = base.band_SDP(n, m, bw, seed=1)
    st = dict(primalstart={"x": P._X0}, dualstart={"y": P._y0, "s": P._S0})
    run("band n=%d m=%d bw=%d feas primal" % (n, m, bw), P, scaling="primal", **st)
    run("band n=%d m=%d bw=%d feas dual  " % (n, m, bw), P, scaling="dual", **st)
    run("band n=%d m=%d bw=%d esd        " % (n, m, bw), P, feas=False)
# 1 x 1 SDP: minimize c x s.t. a x = b, x >= 0
class One(base.SDP):
    def __init__(self):
        super().__init__()
        self._A = sp.csc_matrix(np.array([[2.0, 1.0]]))   # C = 2, A1 = 1
        self._b = np.array([3.0])
        self._blockstruct = [1]
run("1x1 esd", One(), feas=False)

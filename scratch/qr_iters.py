import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from smcp_amd import base, solvers
solvers.options.update(show_progress=False, maxiters=100)
P = base.band_SDP(40, 12, 2, seed=13)
def run(tag):
    for ks in ("chol", "qr"):
        s = P.solve_feas(scaling="primal", kktsolver=ks)
        print(tag, ks, s["status"], s["iterations"], "%.10f" % s["primal objective"], "pres %.1e dres %.1e gap %.1e" % (s["primal infeasibility"], s["dual infeasibility"], s["gap"]))
if torch.cuda.is_available():
    run("device")
else:
    from oracle_backend import oracle_backend
    with oracle_backend():
        run("oracle")

import sys, os, time
sys.path.insert(0, os.getcwd())
from smcp_amd import base, solvers
solvers.options.update(show_progress=False)
P = base.band_SDP(200, 100, 3, seed=0)
for bl in (False, True):
    solvers.options["batched_linesearch"] = bl
    t0 = time.time(); sol = P.solve_feas(scaling="dual", primalstart={"x": P._X0}, dualstart={"y": P._y0, "s": P._S0}); dt = time.time() - t0
    print("GRAPH", os.environ.get("SMCP_PROBE_GRAPH", "1"), "band200 batched", bl, sol["status"], sol["iterations"], "%.3f s" % dt, flush=True)

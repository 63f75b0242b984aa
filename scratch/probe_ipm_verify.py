import sys, os, time
sys.path.insert(0, os.getcwd())
from smcp_amd import base, solvers, chordal
solvers.options.update(show_progress=False, batched_linesearch=True)
P = base.band_SDP(200, 100, 3, seed=0)
orig = chordal.probe_cone
stats = {"calls": 0, "bad": 0}
def checked(b, d, als, kind):
    got = orig(b, d, als, kind)
    want = []
    for al in als:
        T = b + d * al
        try: (chordal.completion if kind == "p" else chordal.cholesky)(T); want.append(True)
        except ArithmeticError: want.append(False)
    stats["calls"] += 1
    if got != want:
        stats["bad"] += 1
        if stats["bad"] <= 6: print("MISMATCH call", stats["calls"], kind, ["%.4g" % a for a in als], got, want, flush=True)
    return want
chordal.probe_cone = checked
sol = P.solve_feas(scaling="dual", primalstart={"x": P._X0}, dualstart={"y": P._y0, "s": P._S0})
print("GRAPH", os.environ.get("SMCP_PROBE_GRAPH", "1"), sol["status"], sol["iterations"], stats, flush=True)

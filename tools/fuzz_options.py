"""Option variants of the feasible-start driver on the device: every variant must reach the same optimum."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, scipy.sparse as sp
from smcp_amd import base, solvers
base_opts = dict(solvers.options)
variants = [dict(), dict(prediction=False), dict(lifting=False), dict(equalsteps=True), dict(amalgamate=False),
            dict(batched_linesearch=False), dict(refinement=0), dict(refinement=3), dict(eta=5.0), dict(tnzcols=0.0),
            dict(tnzcols=1.0), dict(step=0.5), dict(t0=10.0), dict(delta=0.5)]
bad = 0
for case in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    rng = np.random.default_rng(4400 + case)
    objs = []
    for v in variants:
        solvers.options.clear(); solvers.options.update(base_opts); solvers.options.update(show_progress=False, maxiters=200); solvers.options.update(v)
        P = base.band_SDP(int(30 + 25 * case), 8 + 3 * case, 1 + case, seed=77 + case)      # rebuilt: 'amalgamate' acts at construction
        for sc in ("primal", "dual"):
            try:
                s = P.solve_feas(scaling=sc)
                objs.append((str(v), sc, s["status"], s["iterations"], float(s["primal objective"])))
            except Exception as e:
                objs.append((str(v), sc, "EXC " + type(e).__name__ + ": " + str(e)[:80], -1, float("nan")))
    ref = objs[0][4]
    for o in objs:
        ok = o[2] == "optimal" and abs(o[4] - ref) < 2e-5 * (1 + abs(ref))
        if not ok:
            bad += 1
            print("case", case, "BAD", o, "ref", ref, flush=True)
    print("case", case, "variants", len(objs), "iterations", sorted(set(o[3] for o in objs)), flush=True)
print("bad", bad)

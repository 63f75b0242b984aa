"""Max-cut relaxations on random sparse graphs (non-chordal patterns: minimum-degree embedding, column-sparse
constraints -> SCMcolumn2 path) on the device: feasible-start (both scalings, chol and qr) and the embedding driver;
the optimum is checked against a dense eigenvalue bound certificate: X feasible (diag = 1, PSD completable) and
S = C - Diag(y) PSD with <X, S> ~ 0."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from smcp_amd import base, solvers
solvers.options.update(show_progress=False, maxiters=150)
bad = 0
for case in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    rng = np.random.default_rng(6600 + case)
    n = int(rng.integers(20, 160)); ne = int(n * rng.uniform(1.2, 4.0))
    P = base.maxcut_SDP(n, ne, seed=int(rng.integers(1 << 30)))
    out = {}
    for ks in ("chol", "qr"):
        for sc in ("primal", "dual"):
            try:
                s = P.solve_feas(scaling=sc, kktsolver=ks)
                out[(ks, sc)] = (s["status"], s["iterations"], float(s["primal objective"]), s)
            except Exception as e:
                out[(ks, sc)] = ("EXC " + type(e).__name__ + ": " + str(e)[:80], -1, float("nan"), None)
    try:
        s = P.solve_esd()
        out[("chol", "esd")] = (s["status"], s["iterations"], float(s["primal objective"]), s)
    except Exception as e:
        out[("chol", "esd")] = ("EXC " + type(e).__name__ + ": " + str(e)[:80], -1, float("nan"), None)
    objs = [v[2] for v in out.values() if v[0] == "optimal"]
    ok = len(objs) >= 4 and all(v[0] == "optimal" for k, v in out.items() if k[1] != "esd") and max(objs) - min(objs) < 1e-4 * (1 + abs(objs[0]))
    # certificate from one solution
    s = out[("chol", "dual")][3]
    if s is not None and s["status"] == "optimal":
        C = np.asarray(P.get_A(0).todense()); C = C + np.tril(C, -1).T
        S = C - np.diag(s["y"])
        ok = ok and np.linalg.eigvalsh(S).min() > -1e-6 * (1 + abs(objs[0]))
        ok = ok and abs(float(np.sum(s["y"])) - objs[0]) < 1e-4 * (1 + abs(objs[0]))       # dual objective b'y = sum(y)
    print("case", case, "n", n, "edges", ne, {k: v[:3] for k, v in out.items()}, "OK" if ok else "BAD", flush=True)
    bad += not ok
print("bad", bad)

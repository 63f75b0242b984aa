"""Whole interior-point runs on random small problems: kktsolver chol vs qr, feas vs esd, on the device.
Both KKT solvers solve the same systems, so iteration counts and optima must agree."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, scipy.sparse as sp
from smcp_amd import base, solvers, problems
import fuzz_parity
solvers.options.update(show_progress=False, maxiters=150)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
bad = 0
low = lambda M: sp.csc_matrix(sp.tril(M)) if sp.issparse(M) else sp.csc_matrix(np.tril(M))
import signal, time
class _TO(Exception): pass
def _alarm(*a): raise _TO()
signal.signal(signal.SIGALRM, _alarm)
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
for case in range(first, n_cases):
    rng = np.random.default_rng(int(sys.argv[2]) + case if len(sys.argv) > 2 else 5000 + case)
    kind = case % 4
    if kind == 0:
        P = base.band_SDP(int(rng.integers(10, 80)), int(rng.integers(2, 20)), int(rng.integers(0, 5)), seed=int(rng.integers(1 << 30)))
    else:
        pat = fuzz_parity.pattern(rng, [1, 3, 0][kind - 1])
        from smcp_amd.symbolic import Symbolic
        nv = Symbolic(pat).nnz
        m = int(min(rng.integers(2, 16), max(1, nv // 4)))
        P = base.pattern_SDP(pat, m, density=float(rng.choice([0.01, 0.05, 0.2])), seed=int(rng.integers(1 << 30)))
    print("case", case, "kind", kind, "n", P.n, "m", P.m, flush=True)
    starts = dict(primalstart={"x": low(P._X0)}, dualstart={"y": P._y0, "s": low(P._S0)})
    res = {}
    for ks in ("chol", "qr"):
        for sc in ("primal", "dual"):
            try:
                t0 = time.time(); signal.alarm(25)
                s = P.solve_feas(scaling=sc, kktsolver=ks, **starts)
                signal.alarm(0)
                res[(ks, sc)] = (s["status"], s["iterations"], s["primal objective"])
                print("   feas", ks, sc, s["status"], s["iterations"], "%.1f s" % (time.time() - t0), flush=True)
            except BaseException as e:
                signal.alarm(0)
                res[(ks, sc)] = ("EXC " + type(e).__name__ + " " + str(e)[:60], -1, float("nan"))
    try:
        for ks in ("chol", "qr"):
            t0 = time.time(); signal.alarm(25)
            s = P.solve_esd(kktsolver=ks)
            signal.alarm(0)
            res[(ks, "esd")] = (s["status"], s["iterations"], s["primal objective"])
            print("   esd", ks, s["status"], s["iterations"], "%.1f s" % (time.time() - t0), flush=True)
    except BaseException as e:
        signal.alarm(0)
        res[(ks, "esd")] = ("EXC " + type(e).__name__ + " " + str(e)[:60], -1, float("nan"))
    ok = True
    for sc in ("primal", "dual", "esd"):
        a, b = res.get(("chol", sc)), res.get(("qr", sc))
        if a is None or b is None or a[0] != b[0] or abs(a[1] - b[1]) > 1 or not (abs(a[2] - b[2]) <= 1e-5 * (1 + abs(a[2]))):
            ok = False
    objs = [v[2] for v in res.values() if v[0] == "optimal"]
    if objs and max(objs) - min(objs) > 1e-4 * (1 + abs(objs[0])):
        ok = False
    if not ok or any(v[0] != "optimal" for v in res.values()):
        bad += not ok
        print("case", case, "n", P.n, "m", P.m, "OK" if ok else "MISMATCH", res, flush=True)
    if case % 10 == 9:
        print("progress", case + 1, "mismatches", bad, flush=True)
print("cases", n_cases, "mismatches", bad)

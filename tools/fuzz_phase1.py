"""Phase I on random problems on the device: the returned X0 must satisfy <A_i, X0> = b_i and be positive definite
completable; then the feasible-start solver started from it must reach the optimum of the run with the known start."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, scipy.sparse as sp
from smcp_amd import base, solvers
solvers.options.update(show_progress=False, maxiters=150)
bad = 0
for case in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    rng = np.random.default_rng(8800 + case)
    n, m, bw = int(rng.integers(12, 70)), int(rng.integers(2, 14)), int(rng.integers(1, 5))
    P = base.band_SDP(n, m, bw, seed=int(rng.integers(1 << 30)))
    try:
        X0, sol1 = P.solve_phase1()
    except Exception as e:
        print("case", case, "EXC", type(e).__name__, str(e)[:100], flush=True); bad += 1; continue
    if X0 is None:
        print("case", case, "no strictly feasible point found", flush=True); bad += 1; continue
    X0d = np.asarray(X0.todense())
    res = max(abs(np.sum(np.asarray(P.get_A(i + 1).todense()) * X0d) - P.b[i]) for i in range(m)) / (1 + np.abs(P.b).max())
    # positive definite completable: all clique (band) principal blocks positive definite
    mineig = min(np.linalg.eigvalsh(X0d[i:i + bw + 1, i:i + bw + 1]).min() for i in range(n - bw))
    ref = P.solve_feas(primalstart={"x": sp.csc_matrix(np.tril(P._X0))}, dualstart={"y": P._y0, "s": sp.csc_matrix(np.tril(P._S0))})
    s2 = P.solve_feas(primalstart={"x": sp.csc_matrix(sp.tril(X0))})
    ok = res < 1e-8 and mineig > 0 and s2["status"] == "optimal" and abs(s2["primal objective"] - ref["primal objective"]) < 1e-5 * (1 + abs(ref["primal objective"]))
    print("case", case, "n", n, "m", m, "bw", bw, "res %.1e mineig %.2e" % (res, mineig), s2["status"], s2["iterations"], "OK" if ok else "BAD", flush=True)
    bad += not ok
print("bad", bad)

"""Concurrent line-search probes vs sequential factorisations inside whole interior-point runs, random problems."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, scipy.sparse as sp
from smcp_amd import base, solvers, chordal
from smcp_amd.symbolic import Symbolic
import fuzz_parity
solvers.options.update(show_progress=False, batched_linesearch=True, maxiters=100)
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
        if stats["bad"] <= 8: print("MISMATCH", kind, got, want, flush=True)
    return want
chordal.probe_cone = checked
low = lambda M: sp.csc_matrix(sp.tril(M)) if sp.issparse(M) else sp.csc_matrix(np.tril(M))
for case in range(int(sys.argv[1]) if len(sys.argv) > 1 else 18):
    rng = np.random.default_rng(7700 + case)
    pat = fuzz_parity.pattern(rng, case % 6)
    nv = Symbolic(pat).nnz
    m = int(min(rng.integers(2, 16), max(1, nv // 4)))
    P = base.pattern_SDP(pat, m, density=float(rng.choice([0.01, 0.05, 0.2])), seed=int(rng.integers(1 << 30)))
    for sc in ("primal", "dual"):
        s = P.solve_feas(scaling=sc, primalstart={"x": low(P._X0)}, dualstart={"y": P._y0, "s": low(P._S0)})
        print("case", case, "kind", case % 6, "n", P.n, sc, s["status"], s["iterations"], dict(stats), flush=True)
print("TOTAL", stats)

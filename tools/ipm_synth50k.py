# Whole interior-point run (feasible-start driver, dual scaling) on the headline pattern: synth50k, n = 50 000, m = 100,
# strictly feasible by construction (the band_SDP recipe, base.py:600-636, on the nested block-arrow pattern).
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, scipy.sparse as sp, torch
from smcp_amd import chordal, problems, solvers
from smcp_amd.cspmatrix import cspmatrix
from smcp_amd.symbolic import Symbolic
small = "small" in sys.argv[1:]
kktsolver = "qr" if "qr" in sys.argv[1:] else "chol"
pat = problems.nested_block_arrow_pattern(nsub=2, nmid=6) if small else problems.nested_block_arrow_pattern()
n, cp, ri = pat
m = 100
rng = np.random.default_rng(0)
J = np.repeat(np.arange(n), np.diff(cp)); I = ri.astype(np.int64)
nv = len(I)
symb = Symbolic(pat)
def posdef_on_V(seed):
    X = cspmatrix(symb, torch.from_numpy(problems.random_factor_blkval(symb, seed)).cuda())
    chordal.llt(X)                                   # L L^T on V: positive definite with pattern V
    return X.spmatrix(reordered=False, symmetric=False)      # scipy lower triangle, original coordinates
t0 = time.time()
X0, S0 = sp.csc_matrix(posdef_on_V(1)), sp.csc_matrix(posdef_on_V(2))
per = max(1, int(round(0.005 * nv)))
y0 = rng.standard_normal(m); y0 /= np.linalg.norm(y0)
rows, cols, vals = [], [], []
b = np.zeros(m)
x0v = np.asarray(X0[I, J]).ravel()
Csum = np.zeros(nv)
for i in range(m):
    sel = np.sort(rng.choice(nv, size=per, replace=False))
    v = rng.standard_normal(per) / np.sqrt(per)
    rows.append(I[sel] + n * J[sel]); cols.append(np.full(per, i + 1)); vals.append(v)
    w = np.where(I[sel] == J[sel], 1.0, 2.0)
    b[i] = np.sum(w * v * x0v[sel])
    Csum[sel] += y0[i] * v
c = np.asarray(S0[I, J]).ravel() + Csum
rows.insert(0, I + n * J); cols.insert(0, np.zeros(nv, dtype=np.int64)); vals.insert(0, c)
A = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows).astype(np.int64), np.concatenate(cols).astype(np.int64))), shape=(n * n, m + 1))
print("problem built in %.1f s: n=%d |V|=%d m=%d nnz(A_i)=%d" % (time.time() - t0, n, nv, m, per), flush=True)
solvers.options.update(show_progress=True, maxiters=100)
if os.environ.get("VERIFY_PROBES"):           # every probe against the same trials factored one after the other
    orig = chordal.probe_factors
    stats = {"calls": 0, "bad": 0}
    def checked(b, d, als, kind):
        ok, fac = orig(b, d, als, kind)
        want = []
        for al in als:
            T = b + d * al
            try: (chordal.completion if kind == "p" else chordal.cholesky)(T); want.append(True)
            except ArithmeticError: want.append(False)
        stats["calls"] += 1
        if ok != want:
            stats["bad"] += 1
            print("MISMATCH call", stats["calls"], kind, ["%.4g" % a for a in als], ok, want, flush=True)
        return ok, fac
    chordal.probe_factors = checked
if os.environ.get("PROBE_STATS"):             # what the batched line searches ask for and which trial they end up taking
    orig_pf = chordal.probe_factors
    plog = []
    def logged(b, d, als, kind):
        ok, fac = orig_pf(b, d, als, kind)
        plog.append((kind, len(als), "".join("1" if o else "0" for o in ok)))
        return ok, fac
    chordal.probe_factors = logged
t0 = time.time()
if os.environ.get("PROFILE_HOST"):            # host-side view: where the wall time of the driver goes (cProfile, by own time)
    import cProfile, pstats
    pr = cProfile.Profile()
    sol = pr.runcall(solvers.chordalsolver_feas, A, b, primalstart={"x": X0}, dualstart={"y": y0, "s": S0}, scaling="dual", kktsolver=kktsolver)
    dt = time.time() - t0
    pstats.Stats(pr).sort_stats("tottime").print_stats(45)
    pstats.Stats(pr).sort_stats("cumulative").print_stats(60)
else:
    sol = solvers.chordalsolver_feas(A, b, primalstart={"x": X0}, dualstart={"y": y0, "s": S0}, scaling="dual", kktsolver=kktsolver)
    dt = time.time() - t0
if os.environ.get("VERIFY_PROBES"): print("probe check", stats)
if os.environ.get("PROBE_STATS"):
    from collections import Counter
    for key, cnt in sorted(Counter(plog).items()): print("probe", key, "x", cnt)
print("kktsolver", kktsolver, "status", sol["status"], "iterations", sol["iterations"], "pobj %.8g dobj %.8g gap %.2e" % (sol["primal objective"], sol["dual objective"], sol["gap"]),
      "total %.2f s, %.3f s/iteration (incl. symbolic setup)" % (dt, dt / max(1, sol["iterations"])), "dimacs", ["%.1e" % v for v in sol["dimacs"]])

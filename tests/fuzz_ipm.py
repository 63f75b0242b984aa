"""Whole interior-point runs on random small problems (device): both drivers of the reference (feasible-start with primal
and with dual scaling, `solvers.py:49-1327`; the self-dual embedding, `1330-2467`) with both KKT solvers (`kkt_chol`,
`kkt_qr`, `solvers.py:413-541`).  The two KKT solvers solve the same Newton systems, so status, iteration count (+-1) and
optimum must agree between them, and every run of a strictly feasible problem must end `optimal`.  Used by
tests/test_gpu_solvers.py (a few cases) and scratch/fuzz_ipm.py (long runs)."""
import signal
import time

import numpy as np
import scipy.sparse as sp


class _Timeout(Exception):
    pass


def _alarm(*a):
    raise _Timeout()


def run(n_cases, seed0=5000, first=0, verbose=False, limit_s=25):
    """-> list of (case, tag, results) for the cases that disagree or do not end optimal (generator failures excluded)"""
    from smcp_amd import base, solvers
    from smcp_amd.symbolic import Symbolic
    import fuzz_parity
    saved = dict(solvers.options)
    solvers.options.update(show_progress=False, maxiters=150)
    signal.signal(signal.SIGALRM, _alarm)
    low = lambda M: sp.csc_matrix(sp.tril(M)) if sp.issparse(M) else sp.csc_matrix(np.tril(M))
    bad = []
    try:
        for case in range(first, n_cases):
            rng = np.random.default_rng(seed0 + case)
            kind = case % 4
            try:
                if kind == 0:
                    P = base.band_SDP(int(rng.integers(10, 80)), int(rng.integers(2, 20)), int(rng.integers(0, 5)), seed=int(rng.integers(1 << 30)))
                else:
                    pat = fuzz_parity.pattern(rng, [1, 3, 0][kind - 1])
                    nv = Symbolic(pat).nnz
                    m = int(min(rng.integers(2, 16), max(1, nv // 4)))
                    P = base.pattern_SDP(pat, m, density=float(rng.choice([0.01, 0.05, 0.2])), seed=int(rng.integers(1 << 30)))
            except ValueError:
                continue                        # the generator refused (more constraints than nonzeros): not a case
            tag = "case %d kind %d n %d m %d" % (seed0 + case, kind, P.n, P.m)
            if verbose:
                print(tag, flush=True)
            starts = dict(primalstart={"x": low(P._X0)}, dualstart={"y": P._y0, "s": low(P._S0)})
            res = {}

            def attempt(key, fn):
                try:
                    t0 = time.time()
                    signal.alarm(limit_s)
                    s = fn()
                    signal.alarm(0)
                    res[key] = (s["status"], s["iterations"], float(s["primal objective"]))
                    if verbose:
                        print("  ", key, s["status"], s["iterations"], "%.1f s" % (time.time() - t0), flush=True)
                except BaseException as e:      # noqa: a timeout or a solver error is a finding, not a crash of the sweep
                    signal.alarm(0)
                    res[key] = ("EXC " + type(e).__name__ + " " + str(e)[:60], -1, float("nan"))

            for ks in ("chol", "qr"):
                for sc in ("primal", "dual"):
                    attempt((ks, sc), lambda: P.solve_feas(scaling=sc, kktsolver=ks, **starts))
            for ks in ("chol", "qr"):
                attempt((ks, "esd"), lambda: P.solve_esd(kktsolver=ks))
            ok = all(v[0] == "optimal" for v in res.values())
            for sc in ("primal", "dual", "esd"):
                a, b = res[("chol", sc)], res[("qr", sc)]
                if a[0] != b[0] or abs(a[1] - b[1]) > 1 or not (abs(a[2] - b[2]) <= 1e-5 * (1 + abs(a[2]))):
                    ok = False
            objs = [v[2] for v in res.values() if v[0] == "optimal"]
            if objs and max(objs) - min(objs) > 1e-4 * (1 + abs(objs[0])):
                ok = False
            if not ok:
                bad.append((seed0 + case, tag, res))
                if verbose:
                    print("MISMATCH", tag, res, flush=True)
    finally:
        signal.alarm(0)
        solvers.options.clear()
        solvers.options.update(saved)
    return bad

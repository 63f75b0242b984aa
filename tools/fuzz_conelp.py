"""Random LPs (checked against scipy's HiGHS) and mixed cone programs (chol vs qr, KKT conditions recomputed) through
solvers.conelp / lp on the device."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from scipy.optimize import linprog
from smcp_amd import solvers
solvers.options.update(show_progress=False, maxiters=100)
bad = 0
for case in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    rng = np.random.default_rng(300 + case)
    nx, ml = int(rng.integers(2, 12)), int(rng.integers(12, 40))
    G = rng.standard_normal((ml, nx))
    h = G @ rng.standard_normal(nx) + rng.random(ml) + 0.1
    c = -G.T @ (rng.random(ml) + 0.1)
    ref = linprog(c, A_ub=G, b_ub=h, bounds=[(None, None)] * nx, method="highs")
    out = {}
    for ks in ("chol", "qr"):
        s = solvers.lp(c, G, h, kktsolver=ks)
        out[ks] = (s["status"], float(c @ s["x"]) if s["x"] is not None else float("nan"))
    ok = all(v[0] == "optimal" and abs(v[1] - ref.fun) < 1e-5 * (1 + abs(ref.fun)) for v in out.values())
    # mixed cone program: l + q + s blocks, strictly feasible by construction
    nq, ns = int(rng.integers(3, 6)), int(rng.integers(2, 5))
    dims = {"l": int(rng.integers(1, 5)), "q": [nq], "s": [ns]}
    K = dims["l"] + nq + ns * ns
    nxc = min(nx, dims["l"] + 2 * nq - 1 + ns * (ns + 1) // 2 - 1)      # at most as many variables as entries of V
    Gc = rng.standard_normal((K, nxc))
    o = dims["l"] + nq
    for j in range(nxc):                                               # symmetric 's' blocks (CVXOPT convention)
        M = Gc[o:, j].reshape(ns, ns, order="F"); Gc[o:, j] = (0.5 * (M + M.T)).reshape(-1, order="F")
    # a point s0 in the interior of the cone and z0 in the interior of the dual cone
    def interior():
        v = np.zeros(K); v[:dims["l"]] = rng.random(dims["l"]) + 0.5
        q = rng.standard_normal(nq); q[0] = np.linalg.norm(q[1:]) + 1.0; v[dims["l"]:dims["l"] + nq] = q
        M = rng.standard_normal((ns, ns)); M = M @ M.T + ns * np.eye(ns); v[dims["l"] + nq:] = M.reshape(-1, order="F")
        return v
    s0, z0 = interior(), interior()
    hc = Gc @ rng.standard_normal(nxc) + s0
    cc = -Gc.T @ z0
    outc = {}
    for ks in ("chol", "qr"):
        s = solvers.conelp(cc, Gc, hc, dims, kktsolver=ks)
        if s["status"] == "optimal":
            r1 = np.linalg.norm(Gc @ s["x"] + s["s"] - hc) / (1 + np.linalg.norm(hc))
            r2 = np.linalg.norm(Gc.T @ s["z"] + cc) / (1 + np.linalg.norm(cc))
            outc[ks] = ("optimal", float(cc @ s["x"]), r1, r2, abs(float(s["s"] @ s["z"])))
        else:
            outc[ks] = (s["status"], float("nan"), 0, 0, 0)
    okc = all(v[0] == "optimal" and v[2] < 1e-6 and v[3] < 1e-6 for v in outc.values()) and \
        abs(outc["chol"][1] - outc["qr"][1]) < 1e-5 * (1 + abs(outc["chol"][1]))
    if not (ok and okc):
        bad += 1
        print("case", case, "LP", out, "ref", ref.fun, "CONE", outc, flush=True)
    if case % 5 == 4:
        print("progress", case + 1, "bad", bad, flush=True)
print("bad", bad)

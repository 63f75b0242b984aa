#!/usr/bin/env python3
"""Generates tests/golden/ipm_cases.json: whole interior-point runs of the drivers in smcp_amd/solvers.py with every
chordal operation served by the CPU oracle (tests/oracle_backend.py) instead of the HIP library.

What they are: a cross-implementation anchor for the end-to-end path -- the GPU suite must reach the same optimum in
(nearly) the same number of iterations on the same seeded problems (tests/test_gpu_solvers.py), the CPU suite re-runs
them over the oracle (tests/test_host_solvers.py).  What they are NOT: outputs of the reference.

Run from the repo root:  python tests/golden/make_ipm_golden.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = [
    # name, band_SDP(n, m, bw, seed), driver, kwargs
    ("feas_primal", (30, 10, 2, 11), "feas", dict(scaling="primal")),
    ("feas_dual", (50, 15, 3, 12), "feas", dict(scaling="dual")),
    ("feas_qr", (40, 12, 2, 13), "feas", dict(scaling="primal", kktsolver="qr")),
    ("esd_primal", (30, 10, 2, 14), "esd", dict(scaling="primal")),
    ("esd_dual_qr", (36, 9, 3, 15), "esd", dict(scaling="dual", kktsolver="qr")),
    ("feas_lp_like", (25, 8, 0, 16), "feas_started", dict(scaling="dual")),  # bandwidth 0: every clique 1 x 1; known starts
    # patterns with families of small fronts, fronts beyond the LDS class and a deep chain (amalgamated by default)
    ("nested_feas_dual", ("nested", 8, 0.05, 21), "feas_started", dict(scaling="dual")),
    ("arrow_feas_qr", ("arrow", 6, 0.05, 22), "feas_started", dict(scaling="primal", kktsolver="qr")),
    ("chain_esd", (60, 12, 1, 23), "esd", dict(scaling="primal")),
    ("mtxnorm_esd", ("mtxnorm", 6, 4, 24), "esd", dict(scaling="dual")),
]


def build_problem(spec, base):
    from smcp_amd import problems
    if spec[0] == "nested":
        pat = problems.nested_block_arrow_pattern(nsub=2, nmid=3, nleaf_per_mid=4, leaf=(3, 7), mid=(6, 20), top=(24, 40),
                                                  root=70, seed=3)
        return base.pattern_SDP(pat, spec[1], density=spec[2], seed=spec[3])
    if spec[0] == "arrow":
        return base.pattern_SDP(problems.block_arrow_pattern(5, 30, 80), spec[1], density=spec[2], seed=spec[3])
    if spec[0] == "mtxnorm":
        return base.mtxnorm_SDP(spec[1], spec[2], spec[1] + 2, seed=spec[3])
    n, m, bw, seed = spec
    return base.band_SDP(n, m, bw, seed=seed)


def run_case(case, base, solvers):
    name, spec, driver, kw = case
    P = build_problem(spec, base)
    if driver == "feas_started":
        import numpy as np
        import scipy.sparse as sp
        low = lambda M: sp.csc_matrix(sp.tril(M)) if sp.issparse(M) else sp.csc_matrix(np.tril(M))
        kw = dict(kw, primalstart={"x": low(P._X0)}, dualstart={"y": P._y0, "s": low(P._S0)})
    # the stored runs were generated with the reference's ordering sequence (maximum cardinality search): pinned, so that the
    # fixtures stay comparable to 1e-9 whatever the driver's default choice of perfect elimination order is
    peo = solvers.options.get("peo", "auto")
    solvers.options["peo"] = "mcs"
    try:
        sol = (P.solve_esd if driver == "esd" else P.solve_feas)(**kw)
    finally:
        solvers.options["peo"] = peo
    return dict(name=name, status=sol["status"], iterations=int(sol["iterations"]),
                pobj=float(sol["primal objective"]), dobj=float(sol["dual objective"]),
                y=[float(v) for v in sol["y"]])


def main():
    from oracle_backend import oracle_backend
    from smcp_amd import base, solvers
    solvers.options.update(show_progress=False, maxiters=100)
    out = []
    with oracle_backend():
        for case in CASES:
            out.append(run_case(case, base, solvers))
            print(out[-1]["name"], out[-1]["status"], out[-1]["iterations"], out[-1]["pobj"])
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ipm_cases.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Generates tests/golden/phase1_cases.json: SDP.solve_phase1 (reference: src/python/base.py:370-470,
src/C/misc.c:1004-1054) on two seeded problems, one per branch, with every chordal operation served by the CPU oracle:

  leastnorm  max-cut relaxation on a random graph: the least-norm solution of <A_i, X> = b_i (X = I) is positive
             definite completable and is returned as it is (no Phase-I SDP);
  augmented  band SDP whose least-norm solution is NOT in the cone: the Phase-I SDP (one extra 2 x 2 diagonal block,
             trace constraint) is solved by the feasible-start driver and shifted back.

Stored per case: the branch, the Phase-I run (status, iterations, objective), the residual of the equality constraints
at X0, and the main problem solved from X0 (status, objective).  The CPU suite reproduces them over the oracle, the GPU
suite on the device.  A cross-implementation anchor, not an output of the reference.

Run from the repo root:  python tests/golden/make_phase1_golden.py
"""
import json
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = [("leastnorm", ("maxcut", 24, 50, 2)), ("augmented", ("band", 60, 20, 2, 4))]


def build(spec, base):
    if spec[0] == "maxcut":
        return base.maxcut_SDP(spec[1], spec[2], seed=spec[3])
    return base.band_SDP(spec[1], spec[2], spec[3], seed=spec[4])


def run_case(case, base, solvers, chordal):
    name, spec = case
    P = build(spec, base)
    X0, sol1 = P.solve_phase1()
    assert X0 is not None, "Phase I found no strictly feasible point"
    Xd = np.asarray(X0.todense())
    res = max(abs(float(np.sum(np.asarray(P.get_A(i + 1).todense()) * Xd)) - float(P.b[i])) for i in range(P.m))
    Pr = solvers._Problem(P.A, P.b)
    chordal.completion(Pr.from_sym(X0))                       # raises unless X0 is strictly inside the cone
    main = P.solve_feas(scaling="primal", primalstart={"x": sp.csc_matrix(sp.tril(X0))})
    return dict(name=name, branch="least-norm" if sol1 is None else "augmented",
                p1_status=None if sol1 is None else sol1["status"],
                p1_iterations=None if sol1 is None else int(sol1["iterations"]),
                p1_pobj=None if sol1 is None else float(sol1["primal objective"]),
                x0_residual=float(res), status=main["status"], pobj=float(main["primal objective"]))


def main():
    from oracle_backend import oracle_backend
    from smcp_amd import base, chordal, solvers
    solvers.options.update(show_progress=False, maxiters=100)
    out = []
    with oracle_backend():
        for case in CASES:
            out.append(run_case(case, base, solvers, chordal))
            print(out[-1])
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "phase1_cases.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Generates tests/golden/esd_reference_scheme.json: the per-iteration residual history of chordalsolver_esd on a band
SDP (n = 60, m = 20, half-bandwidth 3, seed 1) under the reference's exact refinement scheme
(options esd_kkt_refinement = 0, esd_ds_from_hessian = True: refinement of the outer 5-block Newton system only, dS
through the inverse Hessian, /root/reference/src/python/solvers.py:2017-2056) and under this package's default, with
every chordal operation served by the CPU oracle.

What it documents: the two schemes produce THE SAME iterates down to mu ~ 1e-6 (19 iterations),
i.e. the restatement of bres / tres / newton follows one algorithm; from there on the reference scheme's feasibility
residuals stop improving at 1e-6 .. 1e-7 while its gap keeps shrinking, and it ends 'unknown' at the default
feastol = 1e-8.  Not an output of the reference.

Run from the repo root:  python tests/golden/make_esd_scheme_golden.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PROBLEM = (60, 20, 3, 1)
SCHEMES = {"reference": (0, True), "default": (1, False)}


def run(scheme, base, solvers):
    saved = dict(solvers.options)
    solvers.options.update(show_progress=False, maxiters=60, trace=[])
    solvers.options["esd_kkt_refinement"], solvers.options["esd_ds_from_hessian"] = SCHEMES[scheme]
    try:
        sol = base.band_SDP(*PROBLEM[:3], seed=PROBLEM[3]).solve_esd(scaling="primal")
        trace = list(solvers.options["trace"])
    finally:
        solvers.options.clear()
        solvers.options.update(saved)
    return dict(status=sol["status"], iterations=int(sol["iterations"]), pobj=float(sol["primal objective"]),
                trace=[[int(r[0])] + [float(v) for v in r[1:]] for r in trace])


def main():
    from oracle_backend import oracle_backend
    from smcp_amd import base, solvers
    out = {}
    with oracle_backend():
        for name in SCHEMES:
            out[name] = run(name, base, solvers)
            print(name, out[name]["status"], out[name]["iterations"], out[name]["pobj"])
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "esd_reference_scheme.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()

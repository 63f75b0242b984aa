"""Shared by the CPU and GPU suites: runs the cases of tests/golden/ipm_cases.json and compares with the stored runs."""
import importlib.util
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def _gen():
    spec = importlib.util.spec_from_file_location("make_ipm_golden", os.path.join(HERE, "golden", "make_ipm_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def check_all(iter_slack, obj_tol, y_tol):
    from smcp_amd import base, solvers
    gen = _gen()
    with open(os.path.join(HERE, "golden", "ipm_cases.json")) as f:
        gold = {g["name"]: g for g in json.load(f)}
    assert set(gold) == {c[0] for c in gen.CASES}
    for case in gen.CASES:
        got, g = gen.run_case(case, base, solvers), gold[case[0]]
        assert got["status"] == g["status"] == "optimal", case[0]
        assert abs(got["iterations"] - g["iterations"]) <= iter_slack, (case[0], got["iterations"], g["iterations"])
        assert abs(got["pobj"] - g["pobj"]) <= obj_tol * (1 + abs(g["pobj"])), (case[0], got["pobj"], g["pobj"])
        assert abs(got["dobj"] - g["dobj"]) <= obj_tol * (1 + abs(g["dobj"])), (case[0], got["dobj"], g["dobj"])
        ya, yb = np.array(got["y"]), np.array(g["y"])
        assert np.linalg.norm(ya - yb) <= y_tol * (1 + np.linalg.norm(yb)), case[0]


def check_phase1(iter_slack, obj_tol):
    """Phase-I cases of tests/golden/phase1_cases.json (both branches of SDP.solve_phase1) against the stored runs."""
    from smcp_amd import base, chordal, solvers
    spec = importlib.util.spec_from_file_location("make_phase1_golden", os.path.join(HERE, "golden", "make_phase1_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    with open(os.path.join(HERE, "golden", "phase1_cases.json")) as f:
        gold = {g["name"]: g for g in json.load(f)}
    assert set(gold) == {c[0] for c in gen.CASES}
    for case in gen.CASES:
        got, g = gen.run_case(case, base, solvers, chordal), gold[case[0]]
        assert got["branch"] == g["branch"], case[0]
        assert got["p1_status"] == g["p1_status"], case[0]
        if g["p1_iterations"] is not None:
            assert abs(got["p1_iterations"] - g["p1_iterations"]) <= iter_slack, (case[0], got["p1_iterations"])
            assert abs(got["p1_pobj"] - g["p1_pobj"]) <= 1e-6, (case[0], got["p1_pobj"], g["p1_pobj"])
        assert got["x0_residual"] < 1e-9, (case[0], got["x0_residual"])
        assert got["status"] == g["status"] == "optimal", case[0]
        assert abs(got["pobj"] - g["pobj"]) <= obj_tol * (1 + abs(g["pobj"])), (case[0], got["pobj"], g["pobj"])

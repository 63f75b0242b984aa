import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from smcp_amd import base, chordal, problems, solvers
from smcp_amd.cspmatrix import cspmatrix
from smcp_amd.symbolic import Symbolic
# 1. probes agree with sequential factorisations (both cones), patterns with and without large fronts
for name, pat in (("nested", problems.nested_block_arrow_pattern(nsub=2, nmid=6)), ("arrow_big", problems.block_arrow_pattern(12, 64, 128)), ("band", problems.band_pattern(60, 3))):
    symb = Symbolic(pat); symb.device_init(0, 8)
    X = cspmatrix(symb, torch.from_numpy(problems.random_factor_blkval(symb, 1)).cuda()); chordal.llt(X)
    D = cspmatrix(symb, torch.from_numpy(problems.random_factor_blkval(symb, 2)).cuda()); chordal.llt(D)
    D *= -1.0                                        # X + a D leaves the cone for a large enough
    als = [0.05 * 2 ** k for k in range(8)]
    for kind in ("d", "p"):
        ok = chordal.probe_cone(X, D, als, kind)
        seq = []
        for al in als:
            T = X + D * al
            try:
                (chordal.completion if kind == "p" else chordal.cholesky)(T); seq.append(True)
            except ArithmeticError:
                seq.append(False)
        t0 = time.time()
        for _ in range(5): chordal.probe_cone(X, D, als, kind)
        tb = (time.time() - t0) / 5
        t0 = time.time()
        for _ in range(5):
            for al in als:
                T = X + D * al
                try: (chordal.completion if kind == "p" else chordal.cholesky)(T)
                except ArithmeticError: pass
        ts = (time.time() - t0) / 5
        print(name, kind, "match", ok == seq, ok, "batched %.2f ms vs sequential %.2f ms" % (1e3 * tb, 1e3 * ts), flush=True)
# 2. whole runs
solvers.options.update(show_progress=False)
for bl in (False, True):
    solvers.options["batched_linesearch"] = bl
    P = base.band_SDP(200, 100, 3, seed=0)
    t0 = time.time(); sol = P.solve_feas(scaling="dual", primalstart={"x": P._X0}, dualstart={"y": P._y0, "s": P._S0}); dt = time.time() - t0
    print("band200 batched", bl, sol["status"], sol["iterations"], "%.3f s" % dt, "pobj %.8f" % sol["primal objective"], flush=True)
    P = base.pattern_SDP(problems.nested_block_arrow_pattern(), 100, density=0.005, seed=0)
    t0 = time.time(); sol = P.solve_feas(scaling="dual", primalstart={"x": P._X0}, dualstart={"y": P._y0, "s": P._S0}); dt = time.time() - t0
    print("synth50k batched", bl, sol["status"], sol["iterations"], "%.3f s" % dt, "pobj %.8f" % sol["primal objective"], flush=True)

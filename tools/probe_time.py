"""Time of one probe round (K trial factorisations on the replicated pattern) against K sequential factorisations."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from smcp_amd import chordal, problems
from smcp_amd.cspmatrix import cspmatrix
from smcp_amd.symbolic import Symbolic
symb = Symbolic(problems.nested_block_arrow_pattern())
symb.device_init(0, 1)
X = cspmatrix(symb, torch.from_numpy(problems.random_factor_blkval(symb, 1)).cuda()); chordal.llt(X)
D = cspmatrix(symb, torch.from_numpy(problems.random_factor_blkval(symb, 2)).cuda()); chordal.llt(D); D *= -1.0
for K in (8, 6):
    als = [0.005 * 1.5 ** k for k in range(K)]
    for kind, op in (("d", chordal.cholesky), ("p", chordal.completion)):
        t0 = time.time(); ok = chordal.probe_cone(X, D, als, kind); torch.cuda.synchronize(); t1 = time.time() - t0
        ts = []
        for _ in range(5):
            t0 = time.time(); ok = chordal.probe_cone(X, D, als, kind); torch.cuda.synchronize(); ts.append(time.time() - t0)
        seq = []
        for _ in range(3):
            t0 = time.time()
            want = []
            for al in als:
                T = X + D * al
                try: op(T); want.append(True)
                except ArithmeticError: want.append(False)
            torch.cuda.synchronize(); seq.append(time.time() - t0)
        print("K=%d kind %s: first call %.3f s, probe %.2f ms, sequential %.2f ms, verdicts %s sequential %s" % (K, kind, t1, 1e3 * min(ts), 1e3 * min(seq), ok, want), flush=True)

import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from smcp_amd import chordal, problems
from smcp_amd.cspmatrix import cspmatrix
from smcp_amd.symbolic import Symbolic
for name, pat in (("nested", problems.nested_block_arrow_pattern(nsub=2, nmid=6)), ("band", problems.band_pattern(60, 3))):
    symb = Symbolic(pat); symb.device_init(0, 8)
    X = cspmatrix(symb, torch.from_numpy(problems.random_factor_blkval(symb, 1)).cuda()); chordal.llt(X)
    D = cspmatrix(symb, torch.from_numpy(problems.random_factor_blkval(symb, 2)).cuda()); chordal.llt(D); D *= -1.0
    for kind in ("d", "p"):
        for trial in range(4):
            als = [0.03 * (trial + 1) * 1.7 ** k for k in range(8)]
            ok = chordal.probe_cone(X, D, als, kind)
            seq = []
            for al in als:
                T = X + D * al
                try: (chordal.completion if kind == "p" else chordal.cholesky)(T); seq.append(True)
                except ArithmeticError: seq.append(False)
            print(name, kind, "call", trial, "match", ok == seq, ok, seq, flush=True)

"""Does a completion on another matrix between kkt_qr_factor and solve_ change the solve?"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from smcp_amd import chordal, problems
from smcp_amd.cspmatrix import cspmatrix
from smcp_amd.kkt import KKTSystem
from smcp_amd.symbolic import Symbolic
symb = Symbolic(problems.block_arrow_pattern(5, 30, 80))
symb.device_init(0, 8)
m = 6
cptr, cidx, cval = problems.random_constraints(symb, m, density=0.05, seed=3)
X = cspmatrix(symb, torch.from_numpy(problems.random_factor_blkval(symb, 1)).cuda()); chordal.llt(X)
T0 = cspmatrix(symb, torch.from_numpy(problems.random_factor_blkval(symb, 2)).cuda()); chordal.llt(T0)
msk = np.zeros(symb.blklen, dtype=bool); msk[symb.ccs_to_blk()] = True
b0 = torch.from_numpy(np.random.default_rng(5).standard_normal(symb.blklen) * msk).cuda()
y0 = torch.from_numpy(np.random.default_rng(6).standard_normal(m)).cuda()
for solver in ("qr", "chol"):
    K = KKTSystem(symb, cptr, cidx, cval, max_rhs=8, tnzcols=0.0 if solver == "qr" else None)
    L = X.copy(); chordal.completion(L); Y = X.copy()           # primal scaling: L = completion factor, Y = X
    f = K.factor_qr(L, Y) if solver == "qr" else K.factor(L, Y)
    def solve():
        bx, by = cspmatrix(symb, b0.clone()), y0.clone()
        f(bx, by, 0.5)
        return bx.blkval.clone(), by.clone()
    x1, y1 = solve()
    x1b, y1b = solve()
    for what, op in (("cholesky", chordal.cholesky), ("completion", chordal.completion), ("projected_inverse", chordal.projected_inverse)):
        T = T0.copy()
        if what == "projected_inverse": chordal.cholesky(T)
        op(T)
        x2, y2 = solve()
        print(solver, "after", what, "dx %.2e dy %.2e (repeat: %.2e)" % (float((x2 - x1).abs().max() / x1.abs().max()), float((y2 - y1).abs().max() / y1.abs().max()), float((x1b - x1).abs().max())), flush=True)

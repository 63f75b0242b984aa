# GPU vs oracle search directions at an ill-conditioned scaling point (as near the end of an interior-point run)
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import oracle as orc
from smcp_amd import chordal, problems
from smcp_amd.cspmatrix import cspmatrix
from smcp_amd.kkt import KKTSystem
from smcp_amd.symbolic import Symbolic
symb = Symbolic(problems.nested_block_arrow_pattern(nsub=2, nmid=6, nleaf_per_mid=8, seed=3))
symb.device_init(0, 8)
S = orc.Sym(symb)
msk = np.zeros(symb.blklen, dtype=bool); msk[symb.ccs_to_blk()] = True
for spread in (0, 3, 5, 7):
    rng = np.random.default_rng(10 + spread)
    # L0 with diagonal entries spread over `spread` decades: cond(S) ~ 10^(2 spread)
    Lh = problems.random_factor_blkval(symb, 5)
    nn, na = symb.clique_sizes()
    for k in range(symb.Nsn):
        nf = nn[k] + na[k]
        sc = 10.0 ** (-spread * rng.random(nn[k]))
        blk = Lh[symb.blkptr[k]:symb.blkptr[k] + nf * nn[k]].reshape((nf, nn[k]), order="F")
        blk *= sc[None, :]
    A = Lh.copy(); orc.llt(S, A)
    L = A.copy(); orc.cholesky(S, L); Yh = L.copy(); orc.projected_inverse(S, Yh)
    m = 8
    cptr, cidx, cval = problems.random_constraints(symb, m, density=0.05, seed=9)
    K = orc.KKT(S, cptr, cidx, cval)
    Href = K.schur_factor(L, Yh)
    sys_ = KKTSystem(symb, cptr, cidx, cval, max_rhs=8)
    dev = lambda x: cspmatrix(symb, torch.from_numpy(np.ascontiguousarray(x)).cuda())
    Ld = dev(A); chordal.cholesky(Ld); Yd = Ld.copy(); chordal.projected_inverse(Yd)
    solve = sys_.factor(Ld, Yd)
    bx = rng.standard_normal(symb.blklen) * msk; by = rng.standard_normal(m)
    xr, yr = K.solve(L, Yh, Href, bx, by, 1.0)
    bxd, byd = dev(bx), torch.from_numpy(by.copy()).cuda()
    solve(bxd, byd, 1.0)
    xg, yg = bxd.blkval.cpu().numpy(), byd.cpu().numpy()
    ex = np.linalg.norm((xg - xr)[msk]) / np.linalg.norm(xr[msk]); ey = np.linalg.norm(yg - yr) / np.linalg.norm(yr)
    # residuals of the defining equations, computed by the oracle for both solutions
    def res(x, y):
        r, rr = K.residual(L, Yh, x * msk, y, bx, by, 1.0)
        return np.sqrt(orc.dot(S, r, r)) / max(1, np.sqrt(orc.dot(S, bx, bx))), np.linalg.norm(rr) / max(1, np.linalg.norm(by))
    dl = np.abs(Lh[msk]); 
    print("spread 1e-%d: cond(H)=%.1e  GPU-vs-oracle x %.1e y %.1e | KKT residuals GPU %.1e %.1e oracle %.1e %.1e"
          % (spread, np.linalg.cond(np.tril(Href) @ np.tril(Href).T), ex, ey, *res(xg, yg), *res(xr, yr)), flush=True)

# H of a family pattern with very sparse constraints (the children's fast path) against the CPU oracle
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import oracle as orc
from smcp_amd import chordal, problems
from smcp_amd.cspmatrix import cspmatrix
from smcp_amd.kkt import KKTSystem
from smcp_amd.symbolic import Symbolic
nsub, nmid = int(sys.argv[1]), int(sys.argv[2])
dens = float(sys.argv[3])
symb = Symbolic(problems.nested_block_arrow_pattern(nsub=nsub, nmid=nmid, seed=0))
m = 12
cptr, cidx, cval = problems.random_constraints(symb, m, density=dens, seed=1)
S = orc.Sym(symb)
Lh = problems.random_factor_blkval(symb, 0)
A = Lh.copy(); orc.llt(S, A)
L = A.copy(); orc.cholesky(S, L)
Yh = L.copy(); orc.projected_inverse(S, Yh)
K = orc.KKT(S, cptr, cidx, cval)
Href = K.schur_factor(L, Yh)     # factored
kkt = KKTSystem(symb, cptr, cidx, cval, max_rhs=m, tnzcols=0.0)
Ld = cspmatrix(symb, torch.from_numpy(L).cuda()); Yd = cspmatrix(symb, torch.from_numpy(Yh).cuda())
kkt.build_schur(Ld, Yd, None)
Hraw = np.tril(kkt.H.cpu().numpy().T)
Lr = np.tril(Href)
Hr = Lr @ Lr.T
err = np.abs(np.tril(Hr) - Hraw)
print("cliques", symb.Nsn, "entries/constraint", cptr[1], "max |H| %.3e max err %.3e" % (np.abs(Hr).max(), err.max()))
bad = np.argwhere(err > 1e-9 * np.abs(Hr).max())
print("bad entries", len(bad), bad[:10].tolist())

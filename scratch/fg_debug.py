import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from oracle import oracle as orc
from smcp_amd import problems
from smcp_amd.symbolic import Symbolic
from smcp_amd.kkt import KKTSystem
from test_gpu_parity import _launch_counts, dev, rel
nmid, nleaf, m = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dens = float(sys.argv[4]) if len(sys.argv) > 4 else 0.01
pat = problems.nested_block_arrow_pattern(nsub=1, nmid=nmid, nleaf_per_mid=nleaf, seed=71, shared_mid_sep=True)
symb = Symbolic(pat)
symb.device_init(0, m)
S = orc.Sym(symb)
A = problems.random_factor_blkval(symb, 72)
orc.llt(S, A)
L = A.copy(); orc.cholesky(S, L)
Yh = L.copy(); orc.projected_inverse(S, Yh)
cptr, cidx, cval = problems.random_constraints(symb, m, density=dens, seed=73)
print("nnz per constraint", np.diff(cptr)[:5], "blklen", symb.blklen)
sys_ = KKTSystem(symb, cptr, cidx, cval, max_rhs=m, tnzcols=0.0)
Ld, Yd = dev(symb, L), dev(symb, Yh)
counts = _launch_counts(symb, lambda: sys_.factor(Ld, Yd))
print({k: v for k, v in sorted(counts.items())})
K = orc.KKT(S, cptr, cidx, cval)
Href = K.schur_factor(L, Yh)
print("rel H", rel(np.tril(sys_.H.cpu().numpy().T), np.tril(Href)))

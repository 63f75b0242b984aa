# cycle stamps of k_fam_sparse12 (diagnostic build: SMCP_STAMPS=1 python -m smcp_amd.build --force; run with SMCP_SKIP=64)
import sys, os, ctypes
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from smcp_amd import _lib, chordal, problems
from smcp_amd.cspmatrix import cspmatrix
from smcp_amd.kkt import KKTSystem
from smcp_amd.symbolic import Symbolic
L_ = ctypes.CDLL(_lib.LIB_PATH)
symb = Symbolic(problems.nested_block_arrow_pattern())
m = 100
cptr, cidx, cval = problems.random_constraints(symb, m, density=0.005, seed=1)
kkt = KKTSystem(symb, cptr, cidx, cval, max_rhs=100)
Lh = problems.random_factor_blkval(symb, 0)
L = cspmatrix(symb, torch.from_numpy(Lh).cuda()); S = L.copy(); chordal.llt(S); L = S.copy(); chordal.cholesky(L); Y = L.copy(); chordal.projected_inverse(Y)
kkt.build_schur(L, Y, None)
L_.csp_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
L_.csp_debug_stamps(symb.handle, None, 1)
kkt.build_schur(L, Y, None)
out = (ctypes.c_ulonglong * 32)()
L_.csp_debug_stamps(symb.handle, out, 1)
pn = ['set-up', 'A', 'barrier of the eight', 'B', 'stage barrier']
cn = ['set-up', 'children', 'stage barrier']
for slot, who in ((0, 'wave 0 (tiles)'), (1, 'wave 3 (tiles)'), (2, 'wave 5 (Q0)')):
    print(who, {pn[i]: out[8 * slot + i] // 1792 for i in range(5)})
print('wave 8 (children)', {cn[i]: out[24 + i] // 1792 for i in range(3)})

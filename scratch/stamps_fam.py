# cycle stamps of k_hess_up_fam (diagnostic build: SMCP_STAMPS=1 python -m smcp_amd.build --force; run with SMCP_SKIP=64)
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
pn = ['stage barrier', 'operands + E/X/T', 'U/G/G_NN tiles + stores', 'group barrier', 'Q', 'clear', '-']
cn = ['stage barrier', 'refresh+parent entries', 'children 0-3', 'children 4-7']
tp = sum(out[i] for i in range(7)); tc = sum(out[16 + i] for i in range(4))
print('parent group', {pn[i]: round(100.0 * out[i] / max(tp, 1), 1) for i in range(7)}, 'cycles', tp)
print('child group ', {cn[i]: round(100.0 * out[16 + i] / max(tc, 1), 1) for i in range(4)}, 'cycles', tc)

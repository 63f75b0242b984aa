# cycle stamps of k_fam_sparse (diagnostic build: SMCP_STAMPS=1 python -m smcp_amd.build --force; run with SMCP_SKIP=64)
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
names = ['set-up + fill', 'parent entries', 'wait 1', 'A', 'wait 2', 'B', 'wait 3', 'children']
for rt in range(3):
    tot = sum(out[8 * rt + i] for i in range(8))
    print('role', rt, {names[i]: round(100.0 * out[8 * rt + i] / max(tot, 1), 1) for i in range(8)}, 'cycles per workgroup', tot // 1792)
sn = ['LDS clear + copy issue', 'parent operands', 'child descriptors', 'first barrier', 'counts', 'scan', 'staging', 'drain + barrier']
print('set-up (cycles per workgroup)', {sn[i]: out[24 + i] // 1792 for i in range(8)})

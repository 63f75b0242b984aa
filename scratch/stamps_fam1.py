# cycle stamps of k_hess_up_fam for ONE dense right-hand side (the Hessians of solve_): SMCP_STAMPS=1 python -m smcp_amd.build --force; SMCP_SKIP=64 python3 scratch/stamps_fam1.py
import sys, os, ctypes
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from smcp_amd import _lib, chordal, problems
from smcp_amd.cspmatrix import cspmatrix
from smcp_amd.symbolic import Symbolic
L_ = ctypes.CDLL(_lib.LIB_PATH)
symb = Symbolic(problems.nested_block_arrow_pattern())
symb.device_init(0, 100)
Lh = problems.random_factor_blkval(symb, 0)
L = cspmatrix(symb, torch.from_numpy(Lh).cuda()); S = L.copy(); chordal.llt(S); L = S.copy(); chordal.cholesky(L); Y = L.copy(); chordal.projected_inverse(Y)
U = cspmatrix(symb, torch.randn(symb.blklen, dtype=torch.float64, device='cuda'))
for _ in range(3): chordal.hessian(L, Y, U, adj=False)
L_.csp_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
L_.csp_debug_stamps(symb.handle, None, 1)
N = 10
for _ in range(N): chordal.hessian(L, Y, U, adj=False)
out = (ctypes.c_ulonglong * 32)()
L_.csp_debug_stamps(symb.handle, out, 1)
pn = ['stage barrier', 'operands + E/X/T', 'U/G/G_NN tiles + stores', 'group barrier', 'Q', 'clear', '-']
cn = ['stage barrier', 'refresh+parent entries', 'children 0-3', 'children 4-7']
nwg = 896 * N
print('parent group (cycles per workgroup):', {pn[i]: int(out[i] / nwg) for i in range(7)}, 'total', int(sum(out[i] for i in range(7)) / nwg))
print('child group  (cycles per workgroup):', {cn[i]: int(out[16 + i] / nwg) for i in range(4)}, 'total', int(sum(out[16 + i] for i in range(4)) / nwg))

"""In-kernel wall-clock stamps of k_lf_up3 (workgroup 0, one right-hand side) on synth50k -- diagnostic build only:
SMCP_STAMPS=1 python3 -m smcp_amd.build --force; SMCP_SKIP=64 python3 scratch/stamps_lf.py"""
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
for _ in range(3): chordal.hessian(L, Y, U, adj=None)
L_.csp_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
L_.csp_debug_stamps(symb.handle, None, 1)
N = 20
for _ in range(N): chordal.hessian(L, Y, U, adj=None)
out = (ctypes.c_ulonglong * 32)()
L_.csp_debug_stamps(symb.handle, out, 1)
n = max(1, out[29])
print("k_lf_up3 wg0: calls %d  setup %.2f us  gemm %.2f us  epilogue+drain %.2f us (100 MHz wall clock)" % (n, out[26] / n / 100.0, out[27] / n / 100.0, out[28] / n / 100.0))

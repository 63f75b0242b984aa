import sys, os, ctypes
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from smcp_amd import _lib, chordal, problems
from smcp_amd.cspmatrix import cspmatrix
from smcp_amd.symbolic import Symbolic
L_=ctypes.CDLL(_lib.LIB_PATH)
symb=Symbolic(problems.nested_block_arrow_pattern())
symb.device_init(0,100)
Lh=problems.random_factor_blkval(symb,0)
L=cspmatrix(symb, torch.from_numpy(Lh).cuda()); S=L.copy(); chordal.llt(S); L=S.copy(); chordal.cholesky(L); Y=L.copy(); chordal.projected_inverse(Y)
U=torch.randn(100, symb.blklen, dtype=torch.float64, device='cuda')
chordal.hessian(L,Y,U,adj=False)
L_.csp_debug_stamps.argtypes=[ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
L_.csp_debug_stamps(symb.handle, None, 1)
chordal.hessian(L,Y,U,adj=False)
out=(ctypes.c_ulonglong*32)()
L_.csp_debug_stamps(symb.handle, out, 1)
names=['top-barrier','load/zero','children+scatter','mirror','phase1','phase2','phase3','writeout-panel','writeout-U']
for g,lab in ((0,'leaf'),(16,'mid')):
    tot=sum(out[g+i] for i in range(9))
    print(lab, {names[i]: round(100.0*out[g+i]/max(tot,1),1) for i in range(9)}, 'total', tot)

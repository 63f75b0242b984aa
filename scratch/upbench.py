import sys, os, time, ctypes
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from smcp_amd import _lib, chordal, problems
from smcp_amd.cspmatrix import cspmatrix
from smcp_amd.symbolic import Symbolic
lib=_lib.lib()
symb=Symbolic(problems.nested_block_arrow_pattern())
symb.device_init(0,100)
Lh=problems.random_factor_blkval(symb,0)
L=cspmatrix(symb, torch.from_numpy(Lh).cuda()); S=L.copy(); chordal.llt(S); L=S.copy(); chordal.cholesky(L); Y=L.copy(); chordal.projected_inverse(Y)
U=torch.randn(100, symb.blklen, dtype=torch.float64, device='cuda')
for adj in (False,):
    for it in range(2):
        lib.csp_profile_enable(symb.handle,1); lib.csp_profile_read(symb.handle,None,None)
        chordal.hessian(L,Y,U,adj=adj)
        ms=(ctypes.c_double*64)(); cnt=(ctypes.c_int64*64)(); n=lib.csp_profile_read(symb.handle,ms,cnt)
    print(os.environ.get("SMCP_SKIP"), {lib.csp_profile_kernel_name(i).decode():round(ms[i],2) for i in range(n) if cnt[i]})

import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from smcp_amd import chordal, problems
from smcp_amd.cspmatrix import cspmatrix
from smcp_amd.symbolic import Symbolic
symb=Symbolic(problems.nested_block_arrow_pattern())
symb.device_init(0,100)
Lh=problems.random_factor_blkval(symb,0)
L=cspmatrix(symb, torch.from_numpy(Lh).cuda()); S=L.copy(); chordal.llt(S); L=S.copy(); chordal.cholesky(L); Y=L.copy(); chordal.projected_inverse(Y)
U=torch.randn(100, symb.blklen, dtype=torch.float64, device='cuda')
chordal.hessian(L,Y,U,adj=False)
torch.cuda.synchronize()

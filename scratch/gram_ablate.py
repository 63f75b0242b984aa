# times k_gram_* on synth50k under SMCP_GSKIP ablations (1 = no MFMA phase, 2 = no global loads, 3 = neither)
import sys, os, ctypes
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from smcp_amd import _lib, chordal, problems
from smcp_amd.cspmatrix import cspmatrix
from smcp_amd.kkt import KKTSystem
from smcp_amd.symbolic import Symbolic
lib = _lib.lib()
symb = Symbolic(problems.nested_block_arrow_pattern())
m = 100
cptr, cidx, cval = problems.random_constraints(symb, m, density=0.005, seed=1)
kkt = KKTSystem(symb, cptr, cidx, cval, max_rhs=100)
Lh = problems.random_factor_blkval(symb, 0)
L = cspmatrix(symb, torch.from_numpy(Lh).cuda()); S = L.copy(); chordal.llt(S); L = S.copy(); chordal.cholesky(L); Y = L.copy(); chordal.projected_inverse(Y)
kkt.build_schur(L, Y, None)
h = symb.handle
nk = int(lib.csp_profile_kinds())
names = [lib.csp_profile_kernel_name(i).decode() for i in range(nk)]
lib.csp_profile_filter(h, -1); lib.csp_profile_enable(h, 1); lib.csp_profile_read(h, None, None)
for _ in range(3):
    kkt.build_schur(L, Y, None)
torch.cuda.synchronize()
ms = (ctypes.c_double * nk)(); cnt = (ctypes.c_int64 * nk)()
lib.csp_profile_read(h, ms, cnt)
for i in range(nk):
    if cnt[i] and "gram" in names[i]:
        print("GSKIP", os.environ.get("SMCP_GSKIP", "0"), names[i], "%.3f ms per launch" % (ms[i] / cnt[i]))

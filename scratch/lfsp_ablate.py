"""Wall time of k_lfsp_up on config 3 under SMCP_SKIP bits (32 no Q, 64 no Upd, 128 no G_NN, 256 tables once, 512 no stores)."""
import os, sys, ctypes
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from smcp_amd import chordal, problems, _lib
from smcp_amd.cspmatrix import cspmatrix
from smcp_amd.kkt import KKTSystem
from smcp_amd.symbolic import Symbolic
symb = Symbolic(problems.block_arrow_pattern(2000, 64, 128))
m = 100
cptr, cidx, cval = problems.random_constraints(symb, m, density=0.005, seed=1)
K = KKTSystem(symb, cptr, cidx, cval, max_rhs=60)
S = cspmatrix(symb, torch.from_numpy(problems.random_factor_blkval(symb, 0)).cuda()); chordal.llt(S)
L = S.copy(); chordal.cholesky(L); Y = L.copy(); chordal.projected_inverse(Y)
lib = _lib.lib()
nk = int(lib.csp_profile_kinds()); names = [lib.csp_profile_kernel_name(i).decode() for i in range(nk)]
lib.csp_profile_filter(symb.handle, -1); lib.csp_profile_enable(symb.handle, 1)
for rep in range(2):
    lib.csp_profile_read(symb.handle, None, None)
    try: K.build_schur(L, Y)
    except ArithmeticError: pass
    torch.cuda.synchronize()
    ms = (ctypes.c_double * nk)(); cnt = (ctypes.c_int64 * nk)()
    lib.csp_profile_read(symb.handle, ms, cnt)
print("SMCP_SKIP", os.environ.get("SMCP_SKIP", "0"), {names[i]: round(ms[i], 2) for i in range(nk) if cnt[i] and "lfsp" in names[i]})

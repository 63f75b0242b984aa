"""Per-kernel time of one Hessian application (up + down sweep) on synth50k as a function of the number of
right-hand sides: set-up versus per-pass cost of the small-front kernels.  python tools/hess1_time.py"""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from smcp_amd import _lib, chordal, problems
from smcp_amd.cspmatrix import cspmatrix
from smcp_amd.symbolic import Symbolic
lib = _lib.lib()
symb = Symbolic(problems.nested_block_arrow_pattern())
symb.device_init(0, 16)
S = cspmatrix(symb, torch.from_numpy(problems.random_factor_blkval(symb, 0)).cuda())
chordal.llt(S); L = S.copy(); chordal.cholesky(L); Y = L.copy(); chordal.projected_inverse(Y)
h = symb.handle
nk = int(lib.csp_profile_kinds())
names = [lib.csp_profile_kernel_name(i).decode() for i in range(nk)]
for nr in (1, 2, 4, 8, 16):
    U = torch.randn(nr, symb.blklen, dtype=torch.float64, device="cuda")
    chordal.hessian(L, Y, U, adj=None)
    lib.csp_profile_filter(h, -1); lib.csp_profile_enable(h, 1); lib.csp_profile_read(h, None, None)
    reps = 5
    for _ in range(reps):
        chordal.hessian(L, Y, U, adj=None)
    torch.cuda.synchronize()
    ms = (ctypes.c_double * nk)(); cnt = (ctypes.c_int64 * nk)()
    lib.csp_profile_read(h, ms, cnt); lib.csp_profile_enable(h, 0)
    d = {names[i]: round(1e3 * ms[i] / reps, 1) for i in range(nk) if cnt[i] and ms[i] / reps > 0.01}
    print("nrhs %2d total %.0f us: %s" % (nr, sum(d.values()), d), flush=True)

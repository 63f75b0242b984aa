"""Does the speed of k_fam_terms follow the ALLOCATIONS (physical placement of the stack / exchange buffers) or the process?
One process, the constraint tables and work buffers re-created several times (new device allocations each time; a dummy
allocation of varying size in between shifts what the allocator hands out), k_fam_terms timed by HIP events each time."""
import sys, os, ctypes, gc
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from smcp_amd import _lib, chordal, problems
from smcp_amd.cspmatrix import cspmatrix
from smcp_amd.kkt import KKTSystem
from smcp_amd.symbolic import Symbolic
lib = _lib.lib()
pat = problems.nested_block_arrow_pattern()
m = 100
res = []
keep = []
for trial in range(int(os.environ.get("TRIALS", "6"))):
    symb = Symbolic(pat)
    symb.device_init(0, m)
    cptr, cidx, cval = problems.random_constraints(symb, m, density=0.005, seed=1)
    kkt = KKTSystem(symb, cptr, cidx, cval, max_rhs=m)
    Lh = problems.random_factor_blkval(symb, 0)
    S = cspmatrix(symb, torch.from_numpy(Lh).cuda()); chordal.llt(S)
    L = S.copy(); chordal.cholesky(L); Y = L.copy(); chordal.projected_inverse(Y)
    for _ in range(3): kkt.factor(L, Y)
    h = symb.handle
    nk = int(lib.csp_profile_kinds())
    names = [lib.csp_profile_kernel_name(i).decode() for i in range(nk)]
    lib.csp_profile_filter(h, -1); lib.csp_profile_enable(h, 1); lib.csp_profile_read(h, None, None)
    for _ in range(5): kkt.factor(L, Y)
    torch.cuda.synchronize()
    ms = (ctypes.c_double * nk)(); cnt = (ctypes.c_int64 * nk)()
    lib.csp_profile_read(h, ms, cnt); lib.csp_profile_enable(h, 0)
    i = names.index("k_fam_terms")
    t = ms[i] / max(1, cnt[i])
    res.append(t)
    print("trial %d: k_fam_terms %.3f ms (%d launches)" % (trial, t, cnt[i]), flush=True)
    del kkt, L, Y, S, symb
    gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
    keep.append(torch.empty((trial + 1) * 37_000_001, dtype=torch.float64, device="cuda"))   # shift the next allocations
print("spread", min(res), max(res))

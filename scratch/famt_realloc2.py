"""Which buffer's placement decides the speed of k_fam_terms?  One process, one problem; the packed exchange buffer (updp) or
the constraint stack (ustack) is moved to a fresh allocation (csp_debug_realloc) and k_fam_terms timed after every move."""
import sys, os, ctypes
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from smcp_amd import _lib, chordal, problems
from smcp_amd.cspmatrix import cspmatrix
from smcp_amd.kkt import KKTSystem
from smcp_amd.symbolic import Symbolic
lib = _lib.lib()
L_ = ctypes.CDLL(_lib.LIB_PATH)
L_.csp_debug_realloc.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64]
symb = Symbolic(problems.nested_block_arrow_pattern())
m = 100
symb.device_init(0, m)
cptr, cidx, cval = problems.random_constraints(symb, m, density=0.005, seed=1)
kkt = KKTSystem(symb, cptr, cidx, cval, max_rhs=m)
S = cspmatrix(symb, torch.from_numpy(problems.random_factor_blkval(symb, 0)).cuda()); chordal.llt(S)
L = S.copy(); chordal.cholesky(L); Y = L.copy(); chordal.projected_inverse(Y)
h = symb.handle
nk = int(lib.csp_profile_kinds())
names = [lib.csp_profile_kernel_name(i).decode() for i in range(nk)]
def measure():
    for _ in range(2): kkt.factor(L, Y)
    lib.csp_profile_filter(h, -1); lib.csp_profile_enable(h, 1); lib.csp_profile_read(h, None, None)
    for _ in range(4): kkt.factor(L, Y)
    torch.cuda.synchronize()
    ms = (ctypes.c_double * nk)(); cnt = (ctypes.c_int64 * nk)()
    lib.csp_profile_read(h, ms, cnt); lib.csp_profile_enable(h, 0)
    i = names.index("k_fam_terms"); j = names.index("k_lf_assemble_lds_dyn")
    return ms[i] / max(1, cnt[i]), ms[j] / max(1, cnt[j])
print("start: k_fam_terms %.3f ms  assemble %.3f" % measure(), flush=True)
L_.csp_debug_realloc(h, 20, 0)
for which, tag in ((0, "updp"), (1, "ustack")):
    for t in range(10):
        rc = L_.csp_debug_realloc(h, which, ((t * 7 + 3) % 23) * (32 << 20) + (t % 5) * (2 << 20))
        assert rc == 0, rc
        if os.environ.get("FAST"): print("moved %-6s #%d" % (tag, t), flush=True)
        else: print("moved %-6s #%d: k_fam_terms %.3f ms  assemble %.3f" % ((tag, t) + measure()), flush=True)
        L_.csp_debug_realloc(h, 20, 0)

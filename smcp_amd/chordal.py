"""CHOMPACK-named entry points over the C-ABI (the calls imported at solvers.py:82-97).

Every function mutates its argument in place and raises ``ArithmeticError`` when the matrix is
not positive definite / not PD-completable, exactly as the reference's callers expect
(solvers.py:623-630,638-645).  All arithmetic happens in HIP kernels; there is no CPU path.
"""
import ctypes

import torch

from . import _lib
from .cspmatrix import cspmatrix, _stream, sync_cache


def _chk(rc, what):
    if rc > 0:
        raise ArithmeticError("%s: matrix is not positive definite (clique %d)" % (what, rc - 1))
    if rc < 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, {
            -1: "invalid argument", -2: "no MI355X device initialised (no CPU fallback)",
            -3: "HIP error", -4: "out of memory", -5: "prepared sharded factor overwritten"}.get(rc, "?")))


def _ensure(symb, nrhs=1):
    if symb._device is None or symb._max_rhs < 1:
        if not torch.cuda.is_available():
            raise RuntimeError("smcp_amd needs an MI355X (HIP) device; there is no CPU fallback")
        symb.device_init(torch.cuda.current_device(), max(1, nrhs))


def cholesky(X):
    _ensure(X.symb)
    X.touched()
    _chk(_lib.lib().csp_cholesky(X.symb.handle, X.blkval.data_ptr(), _stream()), "cholesky")


def llt(L):
    _ensure(L.symb)
    L.touched()
    _chk(_lib.lib().csp_llt(L.symb.handle, L.blkval.data_ptr(), _stream()), "llt")


def projected_inverse(L):
    _ensure(L.symb)
    L.touched()
    _chk(_lib.lib().csp_projected_inverse(L.symb.handle, L.blkval.data_ptr(), _stream()), "projected_inverse")


def completion(X):
    _ensure(X.symb)
    X.touched()
    _chk(_lib.lib().csp_completion(X.symb.handle, X.blkval.data_ptr(), _stream()), "completion")


_ADJ = {False: 0, True: 1, None: 2}


def hessian(L, Y, U, adj=False, inv=False):
    """hessian(L, Y, U, adj, inv); U is a cspmatrix, a list of cspmatrices, or a
    (nrhs x ldu) torch tensor whose rows are blkvals (the batched form)."""
    symb = L.symb
    _ensure(symb)
    lib = _lib.lib()
    sync_cache(symb, L, Y)
    a, i = _ADJ[adj], 1 if inv else 0
    if isinstance(U, cspmatrix):
        U.touched()
        _chk(lib.csp_hessian(symb.handle, L.blkval.data_ptr(), Y.blkval.data_ptr(), U.blkval.data_ptr(), 1,
                             symb.blklen, a, i, _stream()), "hessian")
    elif isinstance(U, torch.Tensor):
        assert U.dim() == 2 and U.stride(1) == 1 and U.shape[1] >= symb.blklen
        _chk(lib.csp_hessian(symb.handle, L.blkval.data_ptr(), Y.blkval.data_ptr(), U.data_ptr(), U.shape[0],
                             U.stride(0), a, i, _stream()), "hessian")
    else:
        for Uj in U:
            Uj.touched()
            _chk(lib.csp_hessian(symb.handle, L.blkval.data_ptr(), Y.blkval.data_ptr(), Uj.blkval.data_ptr(), 1,
                                 symb.blklen, a, i, _stream()), "hessian")


def trsm(L, B, trans="N"):
    """B: (n x k) column-major dense right-hand side given as a torch tensor of shape (k, n)
    (row r of the tensor = column r of B), rows in the PERMUTED order."""
    symb = L.symb
    _ensure(symb)
    assert B.dim() == 2 and B.stride(1) == 1 and B.shape[1] == symb.n
    need = -(-int(symb.sepptr[-1]) * B.shape[0] // max(1, 2 * symb.blklen))
    if symb._max_rhs < need:
        symb.device_init(symb._device, need)
    _chk(_lib.lib().csp_trsm(symb.handle, L.blkval.data_ptr(), B.data_ptr(), B.shape[0], B.stride(0),
                             1 if trans in ("T", 1, True) else 0, _stream()), "trsm")


def dot(X, Y):
    _ensure(X.symb)
    out = ctypes.c_double(0.0)
    _chk(_lib.lib().csp_dot(X.symb.handle, X.blkval.data_ptr(), Y.blkval.data_ptr(), ctypes.byref(out), _stream()), "dot")
    return out.value


def logdiagsum(X):
    _ensure(X.symb)
    out = ctypes.c_double(0.0)
    _chk(_lib.lib().csp_logdiagsum(X.symb.handle, X.blkval.data_ptr(), ctypes.byref(out), _stream()), "logdiagsum")
    return out.value


_probe_streams = []


def probe_cone(base, d, alphas, kind):
    """Concurrent trial factorisations (device-resident line search, include/smcp_amd.h csp_probe_*): for every
    alpha in `alphas`, is base + alpha * d inside the cone?  kind 'd': positive definite on V (cholesky, the dual cone
    K_V); kind 'p': positive definite completable (completion, the primal cone C_V).  The K trial matrices are formed
    with one broadcast, each is factored on its own stream with its own workspace slot, and the K failure flags come
    back with one copy -- the reference probes them one after the other (solvers.py:615-689)."""
    import torch
    symb = base.symb
    K = len(alphas)
    _ensure(symb, K)
    if symb._max_rhs < K:                       # slot s uses right-hand-side copy s of the update workspaces
        symb.device_init(symb._device, K)
    L = _lib.lib()
    _chk(L.csp_probe_reserve(symb.handle, K), "csp_probe_reserve")
    al = torch.as_tensor(list(alphas), dtype=torch.float64, device=base.blkval.device)
    # persistent trial buffer per pattern: its address is part of the key of the captured launch sequences
    T = symb.__dict__.get("_probe_T")
    if T is None or T.shape[0] < K or T.device != base.blkval.device:
        T = torch.empty((max(K, 8), symb.blklen), dtype=torch.float64, device=base.blkval.device)
        symb.__dict__["_probe_T"] = T
    Tk = T[:K]
    torch.mul(al.unsqueeze(1), d.blkval.unsqueeze(0), out=Tk)
    Tk.add_(base.blkval.unsqueeze(0))
    out = (ctypes.c_int * K)()
    rc = L.csp_probe_run(symb.handle, 1 if kind == "p" else 0, K, Tk.data_ptr(), T.stride(0), _stream(), out)
    if rc < 0:
        raise RuntimeError("csp_probe_run failed (%d)" % rc)
    return [out[k] == 0 for k in range(K)]

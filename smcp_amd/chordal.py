"""CHOMPACK-named entry points over the C-ABI (the calls imported at solvers.py:82-97).

Every function mutates its argument in place and raises ``ArithmeticError`` when the matrix is
not positive definite / not PD-completable, exactly as the reference's callers expect
(solvers.py:623-630,638-645).  All arithmetic happens in HIP kernels; there is no CPU path.
"""
import ctypes

import torch

from . import _lib
from .cspmatrix import cspmatrix, _stream, note_cache, sync_cache


def _chk(rc, what):
    if rc > 0:
        raise ArithmeticError("%s: matrix is not positive definite (clique %d)" % (what, rc - 1))
    if rc < 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, {
            -1: "invalid argument", -2: "no MI355X device initialised (no CPU fallback)",
            -3: "HIP error", -4: "out of memory",
            -5: "stale derived quantities (prepared sharded factor overwritten, or a matrix changed without csp_touch)",
            -6: "in-launch dependency wait timed out (one-launch blocked Cholesky)"}.get(rc, "?")))


def _ensure(symb, nrhs=1):
    if symb._device is None or symb._max_rhs < 1:
        if not torch.cuda.is_available():
            raise RuntimeError("smcp_amd needs an MI355X (HIP) device; there is no CPU fallback")
        symb.device_init(torch.cuda.current_device(), max(1, nrhs))


def lazy_status(symb, on=True):
    """Deferred failure reports (include/smcp_amd.h: csp_lazy_status): with on=True the factorisations on `symb` no
    longer wait for the device to tell whether the matrix was positive definite -- they return at once and the first
    failure is latched on the device until ``check_status`` reads it.  The reference's calls raise immediately
    (solvers.py:881-891); a driver that knows its scaling point is inside the cone (the line search has just factored
    it) can run a whole KKT solve with ONE host synchronisation this way."""
    _ensure(symb)
    _chk(_lib.lib().csp_lazy_status(symb.handle, 1 if on else 0), "csp_lazy_status")
    symb.__dict__["_lazy_status"] = bool(on)


TUNE_LEAFGRAM, TUNE_VERIFY_CACHE, TUNE_DETERMINISTIC, TUNE_PLACEMENT, TUNE_RACE, TUNE_RACE_DROP_JOINS = 1, 2, 3, 4, 5, 6


def tune(symb, what, value):
    """include/smcp_amd.h: csp_tune -- TUNE_LEAFGRAM (0 never / 1 when cheaper / 2 whenever possible: closed-form Gram
    blocks of childless small cliques), TUNE_VERIFY_CACHE (1: every reuse of a cached derived quantity checks a
    fingerprint of the matrix it came from; a forgotten ``touch`` raises instead of serving stale factors),
    TUNE_DETERMINISTIC (1: fixed-order summation, bit-identical results from run to run), TUNE_PLACEMENT (an action: value =
    tries; after the constraints are set, moves the packed exchange buffer to the fastest of up to `tries` fresh allocations
    for the store pattern of the family sweep -- for runs of many Newton steps on one problem), TUNE_RACE (value = seed,
    0 = off, process-wide: seeded delay injection on every internal stream hand-over, tools/race_hunt.sh)."""
    _ensure(symb)
    _chk(_lib.lib().csp_tune(symb.handle, int(what), int(value)), "csp_tune")


def race_injected(symb):
    """Delay kernels injected so far in this process (csp_tune_report: TUNE_RACE)."""
    import ctypes
    rep = (ctypes.c_double * 3)()
    _chk(_lib.lib().csp_tune_report(symb.handle, rep), "csp_tune_report")
    return int(rep[2])


def touch(X):
    """Tell the library that X.blkval was rewritten by means it cannot see (torch arithmetic on the tensor, a copy into
    it): whatever it had derived from the old contents at that address is dropped (csp_touch).  The wrappers of this
    module track torch's in-place version counter and do this themselves; the call is for code that goes through raw
    pointers, as the reference does with blas.scal(a, X.blkval) (solvers.py:407, 905)."""
    if X.symb._device is not None:
        _chk(_lib.lib().csp_touch(X.symb.handle, X.blkval.data_ptr()), "csp_touch")
    X.touched()


def check_status(symb, what="factorisation (deferred status)"):
    """Synchronises and raises the ArithmeticError a factorisation since the last check would have raised."""
    _chk(_lib.lib().csp_status(symb.handle, _stream()), what)


def cholesky(X):
    _ensure(X.symb)
    X.touched()
    try:
        _chk(_lib.lib().csp_cholesky(X.symb.handle, X.blkval.data_ptr(), _stream()), "cholesky")
    finally:
        note_cache(X.symb, X)


def llt(L):
    _ensure(L.symb)
    L.touched()
    try:
        _chk(_lib.lib().csp_llt(L.symb.handle, L.blkval.data_ptr(), _stream()), "llt")
    finally:
        note_cache(L.symb, L)


def projected_inverse(L):
    _ensure(L.symb)
    L.touched()
    try:
        _chk(_lib.lib().csp_projected_inverse(L.symb.handle, L.blkval.data_ptr(), _stream()), "projected_inverse")
    finally:
        note_cache(L.symb, L)


def cholesky_projected_inverse(L, Y, factors=True):
    """The dual scaling point (solvers.py:881-891): L holds S on entry; on return L = cholesky(S) and Y = projected_inverse(L),
    from ONE library call whose independent stages overlap (csp_cholesky_projected_inverse).  factors: also leave the
    Cholesky factors of the separator blocks of Y behind (what hessian(adj=False / True) and the Schur sweeps use next)."""
    _ensure(L.symb)
    L.touched()
    Y.touched()
    try:
        _chk(_lib.lib().csp_cholesky_projected_inverse(L.symb.handle, L.blkval.data_ptr(), Y.blkval.data_ptr(), 1 if factors else 0,
                                                       _stream()), "cholesky")
    finally:
        note_cache(L.symb, L)
        note_cache(L.symb, Y)


def completion(X):
    _ensure(X.symb)
    X.touched()
    try:
        _chk(_lib.lib().csp_completion(X.symb.handle, X.blkval.data_ptr(), _stream()), "completion")
    finally:
        note_cache(X.symb, X)


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


PROBE_WIDTH = 8       # trial factorisations per probe round (copies of the pattern in the replicated context)


def _forest(symb, K):
    """The K-fold replicated Symbolic of `symb` on the same device, with its persistent trial buffer."""
    import torch
    fs = symb.__dict__.setdefault("_forests", {})
    if K not in fs:
        _ensure(symb)
        F = symb.replicate(K)
        F.device_init(symb._device, 1)
        F.__dict__["_trial_T"] = torch.empty((K, symb.blklen), dtype=torch.float64, device="cuda:%d" % symb._device)
        fs[K] = F
    return fs[K]


def probe_factors(base, d, alphas, kind):
    """Trial factorisations of base + alpha * d for every alpha in `alphas` in ONE factorisation of the K-fold
    replicated pattern (include/smcp_amd.h csp_symbolic_replicate): the launches of a single cholesky (kind 'd': the
    dual cone K_V) or completion (kind 'p': the primal cone C_V), K times as wide, one failure flag per trial.  The
    reference factors its trial points one after the other (solvers.py:615-689, 928-939, 2172-2209).
    Returns (ok, factors): ok[k] = trial k is inside the cone; factors[k] = its factor as a cspmatrix VIEW of the
    trial buffer (valid until the next probe on this pattern), or None where the trial failed."""
    import torch
    symb = base.symb
    K = len(alphas)
    # one replicated pattern per base pattern (building it costs about as much as the base context): rounds of fewer
    # than PROBE_WIDTH trials repeat their last trial, longer lists are split
    if K > PROBE_WIDTH:
        ok, fac = [], []
        for k0 in range(0, K, PROBE_WIDTH):
            o, f = probe_factors(base, d, alphas[k0:k0 + PROBE_WIDTH], kind)
            if k0 + PROBE_WIDTH < K:
                f = [None if x is None else x.copy() for x in f]     # the next round overwrites the trial buffer
            ok += o
            fac += f
        return ok, fac
    alphas = list(alphas) + [alphas[-1]] * (PROBE_WIDTH - K)
    F = _forest(symb, PROBE_WIDTH)
    T = F.__dict__["_trial_T"]
    al = torch.as_tensor(alphas, dtype=torch.float64, device=T.device)
    torch.mul(al.unsqueeze(1), d.blkval.unsqueeze(0), out=T)
    T.add_(base.blkval.unsqueeze(0))
    L = _lib.lib()
    rc = (L.csp_completion if kind == "p" else L.csp_cholesky)(F.handle, T.data_ptr(), _stream())
    if rc < 0:
        _chk(rc, "probe")
    flags = (ctypes.c_int * PROBE_WIDTH)()
    if rc > 0:
        _chk(L.csp_trial_flags(F.handle, PROBE_WIDTH, flags), "csp_trial_flags")
    ok = [flags[k] == 0 for k in range(K)]
    return ok, [cspmatrix(symb, T[k]) if ok[k] else None for k in range(K)]


def probe_cone(base, d, alphas, kind):
    """ok[k]: is base + alphas[k] * d inside the cone?  (probe_factors without the factors.)"""
    return probe_factors(base, d, alphas, kind)[0]

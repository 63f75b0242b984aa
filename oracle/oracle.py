"""ctypes binding of oracle/liboracle.so (chordal_oracle.c) + dense helpers for the checks.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by anything under smcp_amd/.  Parity unpinned (no reference golden
vectors exist for this path, see chordal_oracle.c header); pinned by dense identities instead.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    src = os.path.join(_HERE, "chordal_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


class _CSym(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int64), ("nsn", ctypes.c_int64),
                ("snptr", ctypes.c_void_p), ("snpar", ctypes.c_void_p), ("rowptr", ctypes.c_void_p),
                ("rowidx", ctypes.c_void_p), ("sepptr", ctypes.c_void_p), ("relidx", ctypes.c_void_p),
                ("blkptr", ctypes.c_void_p), ("updptr", ctypes.c_void_p), ("chptr", ctypes.c_void_p),
                ("chidx", ctypes.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB)
        P = ctypes.c_void_p
        S = ctypes.POINTER(_CSym)
        for name in ("orc_cholesky", "orc_llt", "orc_projected_inverse", "orc_completion"):
            getattr(L, name).restype = ctypes.c_int
            getattr(L, name).argtypes = [S, P, P]
        L.orc_hessian.restype = ctypes.c_int
        L.orc_hessian.argtypes = [S, P, P, P, ctypes.c_int, ctypes.c_int, P]
        L.orc_trsm.restype = ctypes.c_int
        L.orc_trsm.argtypes = [S, P, P, ctypes.c_int64, ctypes.c_int64, ctypes.c_int]
        L.orc_dot.restype = ctypes.c_double
        L.orc_dot.argtypes = [S, P, P]
        L.orc_logdiagsum.restype = ctypes.c_double
        L.orc_logdiagsum.argtypes = [S, P]
        L.orc_dense_potrf.restype = ctypes.c_int
        L.orc_dense_potrf.argtypes = [ctypes.c_int64, P, ctypes.c_int64]
        L.orc_dense_potrs.restype = None
        L.orc_dense_potrs.argtypes = [ctypes.c_int64, ctypes.c_int64, P, ctypes.c_int64, P, ctypes.c_int64]
        L.orc_prepare_fac.restype = ctypes.c_int
        L.orc_prepare_fac.argtypes = [S, P, P, P]
        L.orc_hess_g_masked.restype = None
        L.orc_hess_g_masked.argtypes = [S, P, P, P, P, P]
        for name in ("orc_cholesky_masked", "orc_projected_inverse_masked"):
            getattr(L, name).restype = ctypes.c_int
            getattr(L, name).argtypes = [S, P, P, P]
        L.orc_prepare_fac_masked.restype = ctypes.c_int
        L.orc_prepare_fac_masked.argtypes = [S, P, P, P, P]
        L.orc_hess_up_masked.restype = None
        L.orc_hess_up_masked.argtypes = [S, P, P, P, P, P]
        L.orc_hess_down_masked.restype = None
        L.orc_hess_down_masked.argtypes = [S, P, P, P, P]
        L.orc_schur_columns.restype = ctypes.c_int
        L.orc_schur_columns.argtypes = [S, P, P, ctypes.c_int64, P, P, P, P, ctypes.c_int64, ctypes.c_int64, P, ctypes.c_int, P]
        L.orc_set_blas.restype = None
        L.orc_set_blas.argtypes = [P, P, P, P, P, ctypes.c_int]
        L.orc_blas_enabled.restype = ctypes.c_int
        L.orc_scmcolumn2.restype = None
        L.orc_scmcolumn2.argtypes = [ctypes.c_int64, ctypes.c_int64, P, P, P, P, P, P, P, ctypes.c_int64]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def use_blas(on=True, min_dim=32):
    """Route the per-clique dense operations with a dimension >= min_dim through the host BLAS / LAPACK that scipy
    bundles (OpenBLAS; Fortran-interface entry points from scipy.linalg.cython_blas / cython_lapack), one BLAS thread
    (the independent Schur columns are what is spread over OpenMP threads).  Returns a description of the library, or
    None when switched off / unavailable.  The plain loops stay the default: they are the parity checker."""
    L = lib()
    if not on:
        L.orc_set_blas(None, None, None, None, None, int(min_dim))
        return None
    try:
        from scipy.linalg import cython_blas, cython_lapack
    except Exception:
        return None
    ctypes.pythonapi.PyCapsule_GetName.restype = ctypes.c_char_p
    ctypes.pythonapi.PyCapsule_GetName.argtypes = [ctypes.py_object]
    ctypes.pythonapi.PyCapsule_GetPointer.restype = ctypes.c_void_p
    ctypes.pythonapi.PyCapsule_GetPointer.argtypes = [ctypes.py_object, ctypes.c_char_p]

    def fptr(mod, name):
        cap = mod.__pyx_capi__[name]
        return ctypes.pythonapi.PyCapsule_GetPointer(cap, ctypes.pythonapi.PyCapsule_GetName(cap))

    L.orc_set_blas(fptr(cython_blas, "dgemm"), fptr(cython_blas, "dtrsm"), fptr(cython_blas, "dtrmm"),
                   fptr(cython_blas, "dsyrk"), fptr(cython_lapack, "dpotrf"), int(min_dim))
    desc = "scipy-bundled BLAS"
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        threadpool_limits(1, user_api="blas")
        for info in threadpool_info():
            if info.get("user_api") == "blas":
                desc = "%s %s (1 BLAS thread)" % (info.get("internal_api", "blas"), info.get("version", ""))
    except Exception:
        pass
    return desc


class Sym:
    """Symbolic arrays in the shared flat layout.  Build from any object exposing the arrays
    (the product's Symbolic, or oracle.symbolic_ref.symbolic_ref)."""

    FIELDS64 = ("snptr", "snpar", "rowptr", "sepptr", "blkptr", "updptr", "chptr", "chidx")
    FIELDS32 = ("rowidx", "relidx")

    def __init__(self, src):
        get = (lambda k: src[k]) if isinstance(src, dict) else (lambda k: getattr(src, k))
        self.n = int(get("n"))
        for f in self.FIELDS64:
            setattr(self, f, np.ascontiguousarray(get(f), dtype=np.int64))
        for f in self.FIELDS32:
            setattr(self, f, np.ascontiguousarray(get(f), dtype=np.int32))
        self.p = np.ascontiguousarray(get("p"), dtype=np.int64)
        self.nsn = len(self.snptr) - 1
        self.blklen = int(self.blkptr[-1])
        self.updlen = int(self.updptr[-1])
        self.c = _CSym(self.n, self.nsn, *[_p(getattr(self, f)) for f in
                                          ("snptr", "snpar", "rowptr", "rowidx", "sepptr", "relidx",
                                           "blkptr", "updptr", "chptr", "chidx")])
        self._upd = None
        self._work = None

    def ref(self):
        return ctypes.byref(self.c)

    def upd(self):
        if self._upd is None:
            self._upd = np.zeros(max(1, self.updlen))
        return self._upd

    def work(self):
        if self._work is None:
            self._work = np.zeros(max(1, 3 * self.updlen))
        return self._work

    # ---- dense <-> blkval helpers (PERMUTED coordinates) -------------------------------
    def _iter(self):
        for k in range(self.nsn):
            nn = int(self.snptr[k + 1] - self.snptr[k])
            rows = self.rowidx[self.rowptr[k]:self.rowptr[k + 1]].astype(np.int64)
            yield k, nn, rows

    def project(self, M):
        """blkval of P_V(M) for a dense symmetric M in permuted coordinates."""
        out = np.zeros(self.blklen)
        for k, nn, rows in self._iter():
            nf = len(rows)
            blk = M[np.ix_(rows, rows[:nn])].copy()
            blk[:nn, :nn] = np.tril(blk[:nn, :nn])
            out[self.blkptr[k]:self.blkptr[k] + nf * nn] = blk.reshape(-1, order="F")
        return out

    def dense(self, blkval, symmetric=True):
        M = np.zeros((self.n, self.n))
        for k, nn, rows in self._iter():
            nf = len(rows)
            blk = blkval[self.blkptr[k]:self.blkptr[k] + nf * nn].reshape((nf, nn), order="F").copy()
            blk[:nn, :nn] = np.tril(blk[:nn, :nn])
            M[np.ix_(rows, rows[:nn])] = blk
        if symmetric:
            M = M + np.tril(M, -1).T
        return M

    def mask(self):
        """Boolean mask of V (symmetric) in permuted coordinates."""
        M = np.zeros((self.n, self.n), dtype=bool)
        for k, nn, rows in self._iter():
            M[np.ix_(rows, rows[:nn])] = True
        return M | M.T


def _chk(rc, what):
    if rc > 0:
        raise ArithmeticError("%s: not positive definite (clique %d)" % (what, rc - 1))
    if rc < 0:
        raise RuntimeError("%s failed %d" % (what, rc))


def cholesky(S, x):
    _chk(lib().orc_cholesky(S.ref(), _p(x), _p(S.upd())), "cholesky")


def llt(S, x):
    _chk(lib().orc_llt(S.ref(), _p(x), _p(S.upd())), "llt")


def projected_inverse(S, x):
    _chk(lib().orc_projected_inverse(S.ref(), _p(x), _p(S.upd())), "projected_inverse")


def completion(S, x):
    _chk(lib().orc_completion(S.ref(), _p(x), _p(S.upd())), "completion")


_ADJ = {False: 0, True: 1, None: 2}


def hessian(S, L, Y, U, adj=False, inv=False):
    _chk(lib().orc_hessian(S.ref(), _p(L), _p(Y), _p(U), _ADJ[adj], 1 if inv else 0, _p(S.work())), "hessian")


def prepare_fac(S, Y):
    yaa, fac = np.zeros(max(1, S.updlen)), np.zeros(max(1, S.updlen))
    _chk(lib().orc_prepare_fac(S.ref(), _p(Y), _p(yaa), _p(fac)), "prepare_fac")
    return yaa, fac


def hess_g_masked(S, L, fac, u, upd, mask):
    """G sweep (adj=False) over the cliques with mask != 0; upd: this right-hand side's update workspace."""
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    lib().orc_hess_g_masked(S.ref(), _p(L), _p(fac), _p(u), _p(upd), _p(mask))


# ---- masked pieces of the sharded factorisation / solve (tests/test_distributed.py, tests/oracle_backend.py) ----
def _mask(mask):
    return np.ascontiguousarray(mask, dtype=np.uint8)


def cholesky_masked(S, x, upd, mask):
    mask = _mask(mask)
    _chk(lib().orc_cholesky_masked(S.ref(), _p(x), _p(upd), _p(mask)), "cholesky")


def projected_inverse_masked(S, x, upd, mask):
    mask = _mask(mask)
    _chk(lib().orc_projected_inverse_masked(S.ref(), _p(x), _p(upd), _p(mask)), "projected_inverse")


def prepare_fac_masked(S, Y, yaa, fac, mask):
    mask = _mask(mask)
    _chk(lib().orc_prepare_fac_masked(S.ref(), _p(Y), _p(yaa), _p(fac), _p(mask)), "prepare_fac")


def hess_up_masked(S, L, yaa, u, upd, mask):
    mask = _mask(mask)
    lib().orc_hess_up_masked(S.ref(), _p(L), _p(yaa), _p(u), _p(upd), _p(mask))


def hess_down_masked(S, L, u, upd, mask):
    mask = _mask(mask)
    lib().orc_hess_down_masked(S.ref(), _p(L), _p(u), _p(upd), _p(mask))


def trsm(S, L, B, trans="N"):
    """B: (k, n) C-contiguous array = n x k column-major, rows in permuted order."""
    assert B.flags.c_contiguous and B.shape[1] == S.n
    _chk(lib().orc_trsm(S.ref(), _p(L), _p(B), B.shape[0], B.shape[1], 1 if trans == "T" else 0), "trsm")


def dot(S, x, y):
    return lib().orc_dot(S.ref(), _p(x), _p(y))


def logdiagsum(S, x):
    return lib().orc_logdiagsum(S.ref(), _p(x))


def dense_potrf(A):
    """A: (n, n) Fortran-ordered, lower factor in place."""
    assert A.flags.f_contiguous
    _chk(lib().orc_dense_potrf(A.shape[0], _p(A), A.shape[0]), "potrf")


def dense_potrs(A, B):
    assert A.flags.f_contiguous and B.flags.f_contiguous
    nrhs = 1 if B.ndim == 1 else B.shape[1]
    lib().orc_dense_potrs(A.shape[0], nrhs, _p(A), A.shape[0], _p(B), A.shape[0])


# ---- KKT layer restated in numpy around the C kernels (solvers.py:369-386, 477-541) -----
class KKT:
    """Constraints given as CSC over blkval positions: cptr (m+1), cidx, cval (lower-triangle values)."""

    def __init__(self, S, cptr, cidx, cval):
        self.S, self.cptr, self.cidx, self.cval = S, np.asarray(cptr), np.asarray(cidx), np.asarray(cval, dtype=float)
        self.m = len(cptr) - 1
        # diagonal flag of every position
        k = np.searchsorted(S.blkptr, self.cidx, side="right") - 1
        nf = (S.rowptr[k + 1] - S.rowptr[k])
        off = self.cidx - S.blkptr[k]
        self.isdiag = (off // nf) == (off % nf)
        self.w = np.where(self.isdiag, 1.0, 2.0) * self.cval
        self.con = np.repeat(np.arange(self.m), np.diff(self.cptr))

    def amap(self, x):
        return np.bincount(self.con, weights=self.w * x[self.cidx], minlength=self.m)

    def aadj(self, y):
        out = np.zeros(self.S.blklen)
        np.add.at(out, self.cidx, self.cval * y[self.con])
        return out

    def constraint(self, j):
        u = np.zeros(self.S.blklen)
        sl = slice(self.cptr[j], self.cptr[j + 1])
        u[self.cidx[sl]] = self.cval[sl]
        return u

    def schur_factor(self, L, Y, ncols=None):
        """Reference formulation: one Hessian application per constraint (the Python loop at
        solvers.py:479-487), then lapack.potrf (501).  ncols limits the loop (bounded timing sample)."""
        m = self.m
        H = np.zeros((m, m), order="F")
        for j in range(m if ncols is None else ncols):
            u = self.constraint(j)
            hessian(self.S, L, Y, u, adj=None, inv=False)
            H[:, j] = self.amap(u)
        if ncols is None:
            dense_potrf(H)
        return H

    def schur_columns_threaded(self, L, Y, j0, j1, nthreads):
        """Columns j0..j1-1 of the (unfactored) Schur complement on `nthreads` OpenMP threads, one Hessian application
        per column as in schur_factor (orc_schur_columns).  Used by bench.py's cpu_baseline to time the oracle on the
        host's cores."""
        H = np.zeros((self.m, j1 - j0), order="F")
        cptr = np.ascontiguousarray(self.cptr, dtype=np.int64)
        cidx = np.ascontiguousarray(self.cidx, dtype=np.int64)
        cval = np.ascontiguousarray(self.cval, dtype=np.float64)
        w = np.ascontiguousarray(self.w, dtype=np.float64)
        sec = ctypes.c_double(0.0)
        _chk(lib().orc_schur_columns(self.S.ref(), _p(L), _p(Y), self.m, _p(cptr), _p(cidx), _p(cval), _p(w),
                                     int(j0), int(j1), _p(H), int(nthreads), ctypes.byref(sec)), "schur_columns")
        self.last_seconds = sec.value          # compute time without the per-thread workspace allocation
        return H

    def entry_coords(self):
        """(row, col) of every constraint entry in permuted matrix coordinates (row >= col)."""
        S = self.S
        k = np.searchsorted(S.blkptr, self.cidx, side="right") - 1
        nf = S.rowptr[k + 1] - S.rowptr[k]
        off = self.cidx - S.blkptr[k]
        row = S.rowidx[S.rowptr[k] + off % nf]
        col = S.snptr[k] + off // nf
        return np.ascontiguousarray(row, dtype=np.int64), np.ascontiguousarray(col, dtype=np.int64)

    def schur_scm_columns(self, L, H, cols):
        """Columns `cols` of lower(H) by the reference's route for column-sparse constraints (solvers.py:489-497):
        V = unit columns of K_j, trsm(L, V), trsm(L, V, trans='T') -> S^-1[:, K_j], then misc.SCMcolumn2
        (misc.c:620-663, restated as orc_scmcolumn2).  H: (m, m) Fortran-ordered, written in place (rows i >= j)."""
        S, m, n = self.S, self.m, self.S.n
        row, col = self.entry_coords()
        ptr = np.ascontiguousarray(self.cptr, dtype=np.int64)
        val = np.ascontiguousarray(self.cval, dtype=np.float64)
        assert H.flags.f_contiguous and H.shape == (m, m)
        for j in cols:
            sl = slice(ptr[j], ptr[j + 1])
            Kj = np.unique(np.concatenate([row[sl], col[sl]]))
            V = np.zeros((len(Kj), n))
            V[np.arange(len(Kj)), Kj] = 1.0
            trsm(S, L, V, "N")
            trsm(S, L, V, "T")
            kl = np.zeros(n, dtype=np.int64)
            kl[Kj] = np.arange(len(Kj))
            lib().orc_scmcolumn2(m, n, _p(H), _p(ptr), _p(row), _p(col), _p(val), _p(V), _p(kl), int(j))
        return H

    def solve(self, L, Y, H, bx, by, kk):
        r1 = bx.copy()
        hessian(self.S, L, Y, r1, adj=None, inv=False)
        y = np.asfortranarray(kk * by + self.amap(r1))
        dense_potrs(H, y)
        x = self.aadj(y) - bx
        hessian(self.S, L, Y, x, adj=None, inv=False)
        x *= 1.0 / kk
        return x, y

    # ---- kkt_qr (solvers.py:413-475): QR of the stack of half-Hessian images ------------------------------
    def _svec_scale(self):
        """Per blkval position: 0 above the diagonal of a clique's diagonal block (not an entry of the lower
        triangle), 1/sqrt(2) on the diagonal (solvers.py:420), 1 below it."""
        S = self.S
        sc = np.zeros(S.blklen)
        for k, nn, rows in S._iter():
            nf = len(rows)
            i, j = np.meshgrid(np.arange(nf), np.arange(nn), indexing="ij")
            blk = np.where(i > j, 1.0, np.where(i == j, 1.0 / np.sqrt(2.0), 0.0))
            sc[S.blkptr[k]:S.blkptr[k] + nf * nn] = blk.reshape(-1, order="F")
        return sc

    def qr_factor(self, L, Y):
        """kkt_qr's factorisation (solvers.py:414-428): At[:, j] = svec(G(A_j)) with the diagonal entries scaled by
        1/sqrt(2), then lapack.geqrf -- here numpy's Householder QR (LAPACK geqrf + orgqr), thin form."""
        sc = self._svec_scale()
        At = np.zeros((self.S.blklen, self.m))
        for j in range(self.m):
            u = self.constraint(j)
            hessian(self.S, L, Y, u, adj=False, inv=False)
            At[:, j] = sc * u
        Q, R = np.linalg.qr(At, mode="reduced")
        return dict(Q=Q, R=R, sc=sc)

    def qr_solve(self, L, Y, F, bx, by, kk):
        """solve_ of kkt_qr (solvers.py:430-471), statement by statement."""
        import scipy.linalg as sla
        Q, R, sc = F["Q"], F["R"], F["sc"]
        r1 = bx.copy()
        hessian(self.S, L, Y, r1, adj=False, inv=False)
        r1 = sc * r1                                        # spmatrix(...).V, r1[Id] /= sqrt(2)
        x = Q.T @ r1                                        # ormqr(trans='T'), x[m:] = 0
        r2 = sla.solve_triangular(R, by, trans="T")         # trtrs(At[:m,:], r2, uplo='U', trans='T')
        x = x + 0.5 * kk * r2
        y = sla.solve_triangular(R, x)                      # trtrs(At[:m,:], y, uplo='U')
        v = Q @ x - r1                                      # ormqr; Vp.V = x - r1
        # scal_diag(Vp, Id, sqrt(2)) undoes the diagonal scaling: v / sc (off the diagonal sc = 1)
        xo = np.where(sc > 0, v / np.where(sc > 0, sc, 1.0), 0.0)
        hessian(self.S, L, Y, xo, adj=True, inv=False)
        xo *= 1.0 / kk
        return xo, y

    def residual(self, L, Y, x, y, bx, by, kk):
        """kkt_res (solvers.py:401-411): r = -kk*W^-1 x + Aadj(y) - bx ; Amap(x) - by."""
        r = x.copy()
        hessian(self.S, L, Y, r, adj=None, inv=True)
        r *= -kk
        r += self.aadj(y) - bx
        return r, self.amap(x) - by

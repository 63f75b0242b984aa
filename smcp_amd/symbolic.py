"""Host-side symbolic analysis wrapper (counterpart of ``chompack.symbolic``).

Reference call sites: src/python/solvers.py:301-319 (maxcardsearch / peo / symbolic /
sparsity_pattern), analysis.py:49-51,173-175 (supernodes / separators / cliques).
The arithmetic is done by the C-ABI (include/smcp_amd.h, ``csp_symbolic_*``).
"""
import ctypes

import numpy as np

from . import _lib

_Q = dict(scalars=0, p=1, ip=2, snptr=3, snpar=4, rowptr=5, rowidx=6, sepptr=7, relidx=8, blkptr=9,
          updptr=10, chptr=11, chidx=12, levptr=13, levidx=14, ccsptr=15, snode=16)


def _lower_csc(A, n=None):
    """Return (n, colptr, rowind) int64 of the lower-triangular pattern of A.

    A may be a scipy sparse matrix (any triangle / symmetric) or a (n, colptr, rowind) tuple.
    """
    if isinstance(A, tuple):
        n, cp, ri = A
        return int(n), np.ascontiguousarray(cp, dtype=np.int64), np.ascontiguousarray(ri, dtype=np.int64)
    import scipy.sparse as sp
    A = sp.coo_matrix(A)
    n = A.shape[0]
    i = np.maximum(A.row, A.col).astype(np.int64)
    j = np.minimum(A.row, A.col).astype(np.int64)
    key = np.unique(j * n + i)
    j, i = key // n, key % n
    cp = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(j, minlength=n), out=cp[1:])
    return n, cp, np.ascontiguousarray(i)


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def maxcardsearch(A):
    """Maximum cardinality search; returns order with order[new] = orig (solvers.py:301)."""
    n, cp, ri = _lower_csc(A)
    out = np.empty(n, dtype=np.int64)
    rc = _lib.lib().csp_maxcardsearch(n, _ptr(cp), _ptr(ri), _ptr(out))
    if rc:
        raise ValueError("csp_maxcardsearch failed (%d)" % rc)
    return out


def mindegree(A):
    """Greedy minimum-degree ordering (stands in for cvxopt.amd.order, solvers.py:278-279)."""
    n, cp, ri = _lower_csc(A)
    out = np.empty(n, dtype=np.int64)
    rc = _lib.lib().csp_mindegree(n, _ptr(cp), _ptr(ri), _ptr(out))
    if rc:
        raise ValueError("csp_mindegree failed (%d)" % rc)
    return out


class Symbolic:
    """Clique tree + block storage layout of a chordal (embedded) sparsity pattern."""

    def __init__(self, A, p=None):
        n, cp, ri = _lower_csc(A)
        self._cp, self._ri = cp, ri
        perm = None if p is None else np.ascontiguousarray(p, dtype=np.int64)
        info = ctypes.c_int64(0)
        L = _lib.lib()
        self._h = L.csp_symbolic_create(n, _ptr(cp), _ptr(ri), _ptr(perm), ctypes.byref(info))
        if not self._h:
            raise ValueError("csp_symbolic_create failed (%d)" % info.value)
        self._finish()

    def _finish(self):
        sc = self._query("scalars")
        (self.n, self.nnz, self.Nsn, self.fill, self.blklen, self.updlen, self.nlev, self.max_nn,
         self.max_na, self.max_front) = [int(v) for v in sc]
        self._cache = {}
        self._device = None
        self._max_rhs = 0

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().csp_symbolic_destroy(h)
            except Exception:
                pass
            self._h = None

    def replicate(self, K):
        """K independent copies of this pattern as one Symbolic (csp_symbolic_replicate): copy t owns the blkval range
        [t * blklen, (t + 1) * blklen).  One cholesky / completion on it factors K trial matrices at once."""
        info = ctypes.c_int64(0)
        h = _lib.lib().csp_symbolic_replicate(self._h, int(K), ctypes.byref(info))
        if not h:
            raise ValueError("csp_symbolic_replicate failed (%d)" % info.value)
        F = Symbolic.__new__(Symbolic)
        F._cp = F._ri = None
        F._h = h
        F.copies = int(K)
        F._finish()
        return F

    def _query(self, what):
        L = _lib.lib()
        cnt = L.csp_symbolic_query(self._h, _Q[what], None)
        out = np.empty(cnt, dtype=np.int64)
        L.csp_symbolic_query(self._h, _Q[what], _ptr(out))
        return out

    def __getattr__(self, name):
        if name in _Q and name != "scalars":
            c = self.__dict__.setdefault("_cache", {})
            if name not in c:
                c[name] = self._query(name)
            return c[name]
        raise AttributeError(name)

    # ---- chompack-like accessors -------------------------------------------------------
    @property
    def handle(self):
        return self._h

    def is_chordal_input(self):
        return self.fill == 0

    def supernodes(self):
        sp = self.snptr
        return [np.arange(sp[k], sp[k + 1]) for k in range(self.Nsn)]

    def cliques(self):
        rp, ri = self.rowptr, self.rowidx
        return [ri[rp[k]:rp[k + 1]] for k in range(self.Nsn)]

    def separators(self):
        rp, ri, sp = self.rowptr, self.rowidx, self.snptr
        return [ri[rp[k] + (sp[k + 1] - sp[k]):rp[k + 1]] for k in range(self.Nsn)]

    def clique_sizes(self):
        nn = np.diff(self.snptr)
        nf = np.diff(self.rowptr)
        return nn, nf - nn

    def _column_offsets(self):
        """Per column j of supernode k (offset t inside it): clique, t, front size, length of the column in V."""
        sp, rp = self.snptr, self.rowptr
        k = np.repeat(np.arange(self.Nsn), np.diff(sp))
        t = np.arange(self.n) - sp[k]
        nf = (rp[1:] - rp[:-1])[k]
        return k, t, nf, nf - t

    def sparsity_pattern(self):
        """Filled lower pattern in PERMUTED coordinates as (colptr, rowind)."""
        cp = self.ccsptr
        k, t, nf, ln = self._column_offsets()
        start = self.rowptr[k] + t                       # column j lists the clique's rows from its own on
        ri = self.rowidx[np.repeat(start - cp[:-1], ln) + np.arange(self.nnz)].astype(np.int64)
        return cp, ri

    def ccs_to_blk(self):
        """blkval position of every nonzero of sparsity_pattern(), in CCS order."""
        if "ccs2blk" not in self._cache:
            cp = self.ccsptr
            k, t, nf, ln = self._column_offsets()
            base = self.blkptr[k] + t * nf + t           # the diagonal entry of column j inside the clique's panel
            self._cache["ccs2blk"] = np.repeat(base - cp[:-1], ln) + np.arange(self.nnz)
        return self._cache["ccs2blk"]

    def index_map(self, I, J):
        """blkval positions of original-coordinate entries (I, J); -1 where outside V."""
        I = np.ascontiguousarray(I, dtype=np.int64)
        J = np.ascontiguousarray(J, dtype=np.int64)
        out = np.empty(I.shape[0], dtype=np.int64)
        rc = _lib.lib().csp_index_map(self._h, I.shape[0], _ptr(I), _ptr(J), _ptr(out))
        if rc:
            raise ValueError("csp_index_map failed (%d)" % rc)
        return out

    def flops(self):
        """Algorithmic flop / byte counts per SURVEY.md 8(d)."""
        nn, na = self.clique_sizes()
        nn = nn.astype(np.float64)
        na = na.astype(np.float64)
        f_chol = np.sum(nn ** 3 / 3 + na * nn ** 2 + na ** 2 * nn)
        f_pinv = np.sum(2 * nn ** 3 / 3 + 2 * na * nn ** 2 + 2 * na ** 2 * nn)
        f_compl = np.sum(na ** 3 / 3 + 2 * na ** 2 * nn + 2 * na * nn ** 2 + 2 * nn ** 3 / 3)
        f_sweep = np.sum(nn ** 3 + 3 * na * nn ** 2 + 2 * na ** 2 * nn)
        f_h = 2 * f_sweep + np.sum(2 * na ** 2 * nn)
        return dict(chol=f_chol, pinv=f_pinv, completion=f_compl, sweep=f_sweep, hessian=f_h,
                    B=float(self.blklen), U=float(self.updlen))

    # ---- device ------------------------------------------------------------------------
    def device_init(self, device=0, max_rhs=1):
        rc = _lib.lib().csp_device_init(self._h, int(device), int(max_rhs))
        if rc:
            raise RuntimeError("csp_device_init failed (%d): no MI355X visible or out of memory; "
                               "this package has no CPU fallback" % rc)
        self._device = device
        self._max_rhs = max(self._max_rhs, int(max_rhs))
        return self

    def family_roles(self):
        """Per clique, after device_init: 2 = small front swept in one workgroup together with its childless
        children by the family kernel of the sparse-input Schur sweeps (csrc/front_fam.hip), 1 = such a child."""
        L = _lib.lib()
        out = np.zeros(self.Nsn, dtype=np.int64)
        L.csp_symbolic_query(self._h, 17, _ptr(out))
        return out

    def device_bytes(self):
        return int(_lib.lib().csp_device_bytes(self._h))


def amalgamate(symb, nn_max=16, growth=1.3, nf_small=40):
    """Relaxed supernode amalgamation for deep, thin clique trees (chains of tiny cliques: band patterns, the tail
    of a minimum-degree embedding).  Every tree operation costs one kernel launch per level, so a chain of 197
    one-column cliques (config 1: band n = 200) is launch-bound; merging runs of small cliques into supernodes of
    up to `nn_max` columns cuts the depth by the same factor at the price of a few explicit zeros -- the same
    kind of chordal embedding the reference applies to non-chordal patterns (solvers.py:278-319).

    A clique k is merged into its parent p when it is p's LAST child in the postorder (its columns are adjacent to
    p's: no reordering), the merged supernode has at most nn_max columns and the dense block grows by at most
    `growth` (or the merged front has at most nf_small rows).  Returns (pattern, perm) of the embedded pattern --
    pattern as (n, colptr, rowind) in ORIGINAL coordinates, perm[new] = orig -- or None when nothing was merged."""
    nsn = symb.Nsn
    snptr, rowptr, rowidx, par = symb.snptr, symb.rowptr, symb.rowidx, symb.snpar
    nn = np.diff(snptr).astype(np.int64)
    if nsn < 2:
        return None
    # no clique passes the test below against its UNMERGED parent -> no first merge -> nothing to do (the loop only ever
    # creates new candidates by merging): decided without the per-clique row lists (8073 of them on the n = 50 000 benchmark
    # pattern, 40 ms of every problem set-up)
    nf0 = np.diff(rowptr).astype(np.int64)
    nn_new0, nf_new0 = nn[:-1] + nn[1:], nn[:-1] + nf0[1:]
    cand = (np.asarray(par[:-1]) == np.arange(1, nsn)) & (nn_new0 <= nn_max)
    cand &= (nf_new0 * nn_new0 <= growth * (nf0[:-1] * nn[:-1] + nf0[1:] * nn[1:])) | (nf_new0 <= nf_small)
    if not cand.any():
        return None
    first = snptr[:-1].astype(np.int64).copy()          # first permuted column of the (merged) supernode
    rows = [rowidx[rowptr[k]:rowptr[k + 1]].astype(np.int64) for k in range(nsn)]   # front rows, permuted indices
    alive = np.ones(nsn, dtype=bool)
    merged_any = False
    for k in range(nsn - 1):
        p_ = int(par[k])
        if p_ != k + 1:                                  # only the last child: columns adjacent
            continue
        nn_new = int(nn[k] + nn[p_])
        if nn_new > nn_max:
            continue
        nf_k, nf_p = len(rows[k]), len(rows[p_])
        nf_new = int(nn[k]) + nf_p
        if not (nf_new * nn_new <= growth * (nf_k * nn[k] + nf_p * nn[p_]) or nf_new <= nf_small):
            continue
        # merged front: columns of k followed by the whole front of p (A_k is contained in it)
        rows[p_] = np.concatenate([np.arange(first[k], first[k] + nn[k], dtype=np.int64), rows[p_]])
        first[p_] = first[k]
        nn[p_] = nn_new
        alive[k] = False
        merged_any = True
    if not merged_any:
        return None
    perm = np.asarray(symb.p, dtype=np.int64)
    n = symb.n
    ii, jj = [], []
    for k in range(nsn):
        if not alive[k]:
            continue
        r = rows[k]
        c = int(nn[k])                                   # the first c rows of the front are its own columns
        # lower-triangular entries of the dense block [rows x own columns] (permuted indices)
        I = np.repeat(r[:, None], c, axis=1)
        J = np.repeat(r[None, :c], len(r), axis=0)
        m = I >= J
        ii.append(I[m])
        jj.append(J[m])
    I = perm[np.concatenate(ii)]
    J = perm[np.concatenate(jj)]
    lo, hi = np.minimum(I, J), np.maximum(I, J)
    key = np.unique(lo * n + hi)
    cj, ci = key // n, key % n
    cp = np.zeros(n + 1, dtype=np.int64)
    np.add.at(cp, cj + 1, 1)
    return (n, np.cumsum(cp), np.ascontiguousarray(ci)), perm


def symbolic(A, p=None):
    return Symbolic(A, p)

"""Chordal sparse matrix handle: a Symbolic plus a flat fp64 ``blkval`` resident in HBM.

Counterpart of ``chompack.cspmatrix`` as the reference uses it (solvers.py:367,384,407,622):
``X.blkval`` is a flat vector that may be scaled in place, ``copy``, ``+``, ``-``, ``a*X``,
``X.diag()``, ``X.spmatrix()``.  Storage is a torch CUDA (HIP) tensor; torch is plumbing only.
"""
import numpy as np
import torch

from . import _lib


_has_device = []      # torch.cuda.is_available() asks the runtime every time (25 us): once per process


def has_device():
    if not _has_device:
        _has_device.append(torch.cuda.is_available())
    return _has_device[0]


def _stream():
    return torch.cuda.current_stream().cuda_stream if has_device() else None


_uid = [0]


def _cache_reg(symb):
    return symb.__dict__.setdefault("_cache_reg", {})


def note_cache(symb, X):
    """Record the state of X after an in-place library call on it (the library has dropped what it had derived from
    the old contents at this address and may have derived new quantities: csp_projected_inverse leaves the
    inverse-form factor of its input behind for the pair (L, Y = X))."""
    reg = _cache_reg(symb)
    if len(reg) > 256:                     # addresses of long-dead matrices: start over
        if symb._device is not None:
            _lib.lib().csp_cache_reset(symb.handle)
        reg.clear()
    reg[X.blkval.data_ptr()] = X.state()


def sync_cache(symb, *mats):
    """Keep the device-side caches derived from (L, Y) (include/smcp_amd.h, csp_cache_reset) only while they are valid.
    The library keys them by ADDRESS and drops them itself when one of its in-place operations writes to that address;
    what it cannot see is a change made by torch (copy_, +=, ...) or a new matrix in the memory of a dead one.  Every
    matrix that reaches a caching entry point is therefore registered with its state (object identity + torch's
    in-place version counter + the count of in-place library calls); a matrix whose address is registered with a
    different state means the caches may be stale and all of them are dropped."""
    reg = _cache_reg(symb)
    stale = False
    for m in mats:
        st = reg.get(m.blkval.data_ptr())
        if st is not None and st != m.state():
            stale = True
    if stale:
        if symb._device is not None:
            _lib.lib().csp_cache_reset(symb.handle)
        reg.clear()
    for m in mats:
        reg[m.blkval.data_ptr()] = m.state()


class cspmatrix:
    def __init__(self, symb, blkval=None, device=None):
        _uid[0] += 1
        self.uid = _uid[0]
        self._gen = 0
        self.symb = symb
        if blkval is None:
            dev = device if device is not None else ("cuda:%d" % symb._device if symb._device is not None else "cpu")
            blkval = torch.zeros(symb.blklen, dtype=torch.float64, device=dev)
        self.blkval = blkval

    def state(self):
        return (self.uid, self._gen, self.blkval.data_ptr(), self.blkval._version)

    def touched(self):
        """Call after the library wrote into blkval through its raw pointer."""
        self._gen += 1

    # ---- construction ------------------------------------------------------------------
    @classmethod
    def from_entries(cls, symb, I, J, V, device=None):
        """Symmetric matrix given by lower- (or upper-) triangular entries in ORIGINAL coordinates."""
        pos = symb.index_map(I, J)
        if (pos < 0).any():
            raise ValueError("entry outside the sparsity pattern")
        h = np.bincount(pos, weights=np.asarray(V, dtype=np.float64), minlength=symb.blklen).astype(np.float64, copy=False)     # repeated entries add up
        X = cls(symb, device=device)
        X.blkval.copy_(torch.from_numpy(h))
        return X

    @classmethod
    def from_scipy(cls, symb, A, device=None):
        import scipy.sparse as sp
        A = sp.tril(sp.coo_matrix(A)).tocoo()
        return cls.from_entries(symb, A.row, A.col, A.data, device=device)

    @classmethod
    def from_dense_projection(cls, symb, M, device=None):
        """P_V(M) for a dense symmetric matrix M in ORIGINAL coordinates."""
        cp, ri = symb.sparsity_pattern()
        cols = np.repeat(np.arange(symb.n), np.diff(cp))
        p = symb.p
        vals = np.asarray(M)[p[ri], p[cols]]
        h = np.zeros(symb.blklen)
        h[symb.ccs_to_blk()] = vals
        X = cls(symb, device=device)
        X.blkval.copy_(torch.from_numpy(h))
        return X

    # ---- conversions -------------------------------------------------------------------
    def to_dense(self, reordered=False):
        """Dense symmetric numpy matrix (ORIGINAL coordinates unless reordered=True)."""
        s = self.symb
        cp, ri = s.sparsity_pattern()
        cols = np.repeat(np.arange(s.n), np.diff(cp))
        v = self.blkval.detach().cpu().numpy()[s.ccs_to_blk()]
        M = np.zeros((s.n, s.n))
        M[ri, cols] = v
        M[cols, ri] = v
        if not reordered:
            ip = s.ip
            M = M[np.ix_(ip, ip)]
        return M

    def to_dense_factor(self):
        """Dense lower-triangular factor in PERMUTED coordinates (for L produced by cholesky/completion)."""
        s = self.symb
        cp, ri = s.sparsity_pattern()
        cols = np.repeat(np.arange(s.n), np.diff(cp))
        v = self.blkval.detach().cpu().numpy()[s.ccs_to_blk()]
        M = np.zeros((s.n, s.n))
        M[ri, cols] = v
        return M

    def spmatrix(self, reordered=True, symmetric=False):
        """scipy CSC of the lower triangle (permuted coordinates), like X.spmatrix(...) at solvers.py:370."""
        import scipy.sparse as sp
        s = self.symb
        v = self.blkval.detach().cpu().numpy()[s.ccs_to_blk()]
        # The symmetrisation and the un-permutation move VALUES around a pattern that depends on the symbolic factorisation
        # only: done once per (symbolic, form) on a matrix of entry numbers, every later export is one gather (the two exports
        # at the end of an interior-point run on the n = 50 000 benchmark pattern were 0.18 s of its 0.97 s).
        plans = s._cache.setdefault("export_plans", {})
        key = (bool(reordered), bool(symmetric))
        if key not in plans:
            cp, ri = s.sparsity_pattern()
            L = sp.csc_matrix((np.arange(1, len(ri) + 1, dtype=np.float64), ri, cp), shape=(s.n, s.n))   # (entry number + 1: exact in fp64)
            if symmetric:
                L = L + sp.tril(L, -1).T
            if not reordered:
                ip = s.ip
                L = sp.csc_matrix(L.tocsr()[ip][:, ip])
            L.sort_indices()
            plans[key] = (L.indptr.copy(), L.indices.copy(), np.rint(L.data).astype(np.int64) - 1)
        indptr, indices, src = plans[key]
        return sp.csc_matrix((v[src], indices.copy(), indptr.copy()), shape=(s.n, s.n))

    def diag(self):
        s = self.symb
        nn, na = s.clique_sizes()
        nf = nn + na
        idx = np.concatenate([s.blkptr[k] + np.arange(nn[k]) * (nf[k] + 1) for k in range(s.Nsn)])
        return self.blkval[torch.as_tensor(idx, device=self.blkval.device)]

    # ---- arithmetic (flat, on blkval) --------------------------------------------------
    def copy(self):
        return cspmatrix(self.symb, self.blkval.clone())

    def _axpby(self, a, x, b):
        rc = _lib.lib().csp_axpby(self.symb.blklen, a, x.blkval.data_ptr() if x is not None else None, b,
                                  self.blkval.data_ptr(), _stream())
        if rc:
            raise RuntimeError("csp_axpby failed (%d)" % rc)
        self.touched()
        return self

    def __add__(self, other):
        return cspmatrix(self.symb, self.blkval + other.blkval)

    def __sub__(self, other):
        return cspmatrix(self.symb, self.blkval - other.blkval)

    def __iadd__(self, other):
        self.blkval += other.blkval
        return self

    def __isub__(self, other):
        self.blkval -= other.blkval
        return self

    def __mul__(self, a):
        return cspmatrix(self.symb, self.blkval * float(a))

    __rmul__ = __mul__

    def __imul__(self, a):
        self.blkval *= float(a)
        return self

    def __neg__(self):
        return cspmatrix(self.symb, -self.blkval)

    def __pos__(self):
        return self.copy()

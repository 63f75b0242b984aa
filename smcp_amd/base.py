"""Problem container + generators + SDPA-sparse I/O (host side).

Counterparts of the reference's ``SDP`` (src/python/base.py:35-511), ``band_SDP`` (563-636) and the
SDPA reader/writer in src/C/misc.c:139-365 (format: comment lines, m, nBlocks, block sizes with
negative = diagonal block, b, then ``matno blkno i j val`` 1-based upper-triangular entries).
The reference draws random data from cvxopt's RNG; these generators use numpy ``default_rng``.
"""
import math
import os
import re

import numpy as np
import scipy.sparse as sp


class SDP:
    """minimize <C,X> s.t. <A_i,X> = b_i, X psd-completable; data as an n^2 x (m+1) sparse matrix
    with columns vec(C), vec(A_1)... (lower triangles), exactly the reference's layout."""

    def __init__(self, filename=None, c=None, G=None, h=None, dims=None):
        """SDP(fname) reads a problem file.  The reference's constructor also names c, G, h, dims (base.py:62) and then
        ignores them -- its conelp fills the object in by hand (solvers.py:2493-2534); here SDP(c=, G=, h=, dims=) builds the
        block-diagonal SDP whose DUAL is the cone program  minimize c'x s.t. Gx + s = h, s in K, exactly as conelp does:
        column 0 = the embedded h, column j = the embedded j-th column of G, b = -c."""
        self._A = None
        self._b = None
        self._blockstruct = None
        self._pname = None
        self._X0 = self._y0 = self._S0 = None
        if filename is not None:
            # extension dispatch of the reference (base.py:71-86): .dat-s[.bz2] -> SDPA sparse, .pkl[.bz2] -> pickle
            filename = str(filename)
            fp, ext = os.path.splitext(filename)
            self._pname = fp.split("/")[-1]
            if ext == ".dat-s":
                self._read_sdpa(filename)
            elif ext == ".pkl":
                self._load(filename)
            elif os.path.splitext(fp)[1] == ".dat-s":
                self._pname = self._pname[:-6]
                self._read_sdpa(filename)
            elif os.path.splitext(fp)[1] == ".pkl":
                self._pname = self._pname[:-4]
                self._load(filename)
            else:
                raise NameError("Unknown file extension")
        elif c is not None or G is not None or h is not None:
            if c is None or G is None or h is None:
                raise ValueError("SDP(c=, G=, h=, dims=): c, G and h are all needed")
            from .solvers import _embed_columns
            G = sp.csc_matrix(G)
            if dims is None:
                dims = {"l": G.shape[0], "q": [], "s": []}
            hh = np.asarray(h, dtype=np.float64).reshape(-1, 1)
            A, n, Nl, Nq, Ns = _embed_columns(sp.hstack([sp.csc_matrix(hh), G]), dims)
            self._A = sp.csc_matrix(A)
            self._b = -np.asarray(c, dtype=np.float64).reshape(-1, 1)
            self._blockstruct = [-int(Nl)] * (1 if Nl else 0) + [int(q) for q in Nq] + [int(s_) for s_ in Ns]
            self._pname = "conelp"

    def __str__(self):
        return "<SDP: n=%i, m=%i, nnz=%i> %s" % (self.n, self.m, self.nnz, self._pname)

    def _read_sdpa(self, fname):
        """SDPA sparse file -> (A, b, blockstruct), NEGATED as the reference does (base.py:177-194, neg=True):
        an SDPA file states  max <F0,Y> s.t. <Fi,Y> = ci, so C = -F0, A_i = -F_i, b = -c."""
        fp, ext = os.path.splitext(fname)
        if ext == ".bz2":
            import bz2
            import tempfile
            with open(fname, "rb") as fc, tempfile.NamedTemporaryFile("wb", suffix=".dat-s", delete=False) as fo:
                fo.write(bz2.decompress(fc.read()))
                tmp = fo.name
            try:
                self._A, self._b, self._blockstruct = sdpa_read(tmp, neg=True)
            finally:
                os.remove(tmp)
        else:
            self._A, self._b, self._blockstruct = sdpa_read(fname, neg=True)

    def _load(self, fname):
        """Load SDP data from a pickle written by save() (base.py:219-239; binary modes, which the reference's
        text-mode open() breaks on Python 3)."""
        import pickle
        fp, ext = os.path.splitext(fname)
        if ext == ".bz2":
            import bz2
            with open(fname, "rb") as f:
                D = pickle.loads(bz2.decompress(f.read()))
        elif ext == ".pkl":
            with open(fname, "rb") as f:
                D = pickle.load(f)
        else:
            raise IOError("unknown extension '%s' " % ext)
        self._A, self._b = D["A"], D["b"]
        self._X0, self._y0, self._S0 = D["X0"], D["y0"], D["S0"]
        self._pname = D["pname"]
        self._blockstruct = D.get("blockstruct")

    def save(self, fname=None, compress=False):
        """Save SDP data to a pickle file (base.py:241-270): fname + '.pkl' (or '.pkl.bz2')."""
        import pickle
        if fname is None:
            fname = self._pname
        fname += ".pkl.bz2" if compress else ".pkl"
        if os.path.isfile(fname):
            raise IOError("file %s already exists" % fname)
        D = {"A": self._A, "b": self._b, "X0": self._X0, "y0": self._y0, "S0": self._S0, "pname": self._pname,
             "blockstruct": self._blockstruct}
        with open(fname, "wb") as f:
            if compress:
                import bz2
                f.write(bz2.compress(pickle.dumps(D)))
            else:
                pickle.dump(D, f)

    # ---- properties (base.py:92-314) ---------------------------------------------------
    @property
    def n(self):
        return int(round(math.sqrt(self._A.shape[0])))

    @property
    def m(self):
        return self._A.shape[1] - 1

    @property
    def A(self):
        return self._A

    @property
    def b(self):
        return self._b

    @property
    def blockstruct(self):
        return self._blockstruct

    def _require(self):
        if self._A is None:
            raise AttributeError("SDP object has not been initialized")

    @property
    def I(self):
        """Aggregate sparsity pattern as absolute indices into vec(X), lower triangle (base.py:110-117; sorted here -- the
        reference's order is that of a Python set)."""
        self._require()
        return np.unique(sp.csc_matrix(self._A).indices).astype(np.int64)

    @property
    def nnz(self):
        """Number of nonzeros in the lower triangle of the aggregate sparsity pattern."""
        return len(self.I)

    @property
    def issparse(self):
        """True if the aggregate sparsity density is at most one half (base.py:128-131)."""
        return len(self.I) <= 0.5 * (self.n * (self.n + 1) / 2)

    def get_nnz(self, i=None):
        """Nonzeros in the lower triangles of A_0 (= C), A_1, ..., A_m: the whole vector, or entry i (base.py:279-294)."""
        self._require()
        cnt = np.diff(sp.csc_matrix(self._A).indptr).astype(np.int64)
        if i is None:
            return cnt
        if 0 <= i <= self.m:
            return int(cnt[i])
        raise ValueError("index out of range")

    def get_nzcols(self, i=None):
        """Number of nonzero columns (= rows: the matrices are symmetric) of A_1, ..., A_m -- misc.nzcolumns (misc.c:682-730):
        the distinct row and column indices the stored entries touch; the whole vector, or the figure of A_i, 1 <= i <= m
        (base.py:300-310)."""
        self._require()
        A = sp.csc_matrix(self._A)
        n = self.n
        out = np.zeros(self.m, dtype=np.int64)
        for j in range(self.m):
            idx = A.indices[A.indptr[j + 1]:A.indptr[j + 2]]
            out[j] = len(np.union1d(idx % n, idx // n))
        if i is None:
            return out
        if 0 < i <= self.m:
            return int(out[i - 1])
        raise ValueError("index out of range")

    nzcols = property(get_nzcols, doc="Vector with number of nonzero columns in A1,..,Am")

    @property
    def V(self):
        """Aggregate sparsity pattern (lower triangle) as a scipy matrix of ones."""
        n = self.n
        idx = np.unique(sp.csc_matrix(self._A).indices)
        return sp.csc_matrix((np.ones(len(idx)), (idx % n, idx // n)), shape=(n, n))

    nnzs = property(get_nnz, doc="Vector with number of nonzeros in lower triangle of A0,A1,...,Am")

    @property
    def ischordal(self):
        from .symbolic import Symbolic, maxcardsearch
        V = self.V + sp.identity(self.n)
        return Symbolic(V, maxcardsearch(V)).fill == 0

    def get_A(self, i):
        """A_i (i = 0 is C) as a symmetric scipy matrix."""
        n = self.n
        col = sp.csc_matrix(self._A[:, i]).tocoo()
        L = sp.csc_matrix((col.data, (col.row % n, col.row // n)), shape=(n, n))
        return L + sp.tril(L, -1).T

    # ---- solvers (base.py:316-368) -----------------------------------------------------
    def solve_esd(self, kktsolver="chol", scaling="primal", primalstart=None, dualstart=None):
        from . import solvers
        return solvers.chordalsolver_esd(self._A, self._b, primalstart, dualstart, scaling=scaling,
                                         kktsolver=kktsolver)

    def solve_feas(self, kktsolver="chol", scaling="primal", primalstart=None, dualstart=None):
        from . import solvers
        return solvers.chordalsolver_feas(self._A, self._b, primalstart, dualstart, scaling=scaling,
                                          kktsolver=kktsolver)

    def solve_phase1(self, kktsolver="chol", MM=1e5):
        """Primal Phase I with the feasible-start solver (reference: base.py:370-470, misc.phase1_sdp
        misc.c:1004-1054).  Returns (X0, sol): a primal strictly feasible X0 (symmetric scipy matrix) and the
        Phase-I solution dict (None when the least-norm solution of <A_i, X> = b_i is already feasible), or
        (None, P1) with the Phase-I SDP object when no strictly feasible point was found."""
        import scipy.sparse.linalg as spla

        from . import chordal, solvers
        n, m = self.n, self.m
        k = 1e-3
        A = sp.csc_matrix(self._A)
        Id = np.arange(n) * (n + 1)
        b = np.asarray(self._b, dtype=np.float64).reshape(-1)
        # least-norm solution X = 1/2 sum_i u_i A_i with (As^T As) u = b, As = A scaled by 1/sqrt 2 on the diagonal rows
        scale = np.ones(n * n)
        scale[Id] = 1.0 / math.sqrt(2.0)
        As = sp.diags(scale) @ A[:, 1:]
        Mm = (As.T @ As).tocsc()
        u = spla.spsolve(Mm, b) if m > 1 else b / Mm[0, 0]
        x = 0.5 * (A[:, 1:] @ u)
        V = self.V.tocoo()
        X0 = sp.csc_matrix((x[V.row + V.col * n], (V.row, V.col)), shape=(n, n))     # lower triangle on the pattern
        P = solvers._Problem(A, b)

        def feasible(Xlow):
            Xc = P.from_sym(Xlow)
            try:
                chordal.completion(Xc)
                return True
            except ArithmeticError:
                return False

        sym = lambda L_: L_ + sp.tril(L_, -1).T
        if feasible(X0):
            return sym(X0), None
        # Phase-I SDP:  minimize  X'[n,n]  s.t.  <A_i, X> - tr(A_i) X'[n,n] = b_i - k tr(A_i),
        #                                       tr(X) + X'[n+1,n+1] = MM,   X' = blkdiag(X, x_n, x_{n+1}) psd
        trA = np.asarray(A[Id, 1:].sum(axis=0)).reshape(-1)
        n2 = n + 2
        rows, cols, vals = [n * n2 + n], [0], [1.0]
        Ac = A[:, 1:].tocoo()
        rows += list(Ac.row + 2 * (Ac.row // n)); cols += list(Ac.col + 1); vals += list(Ac.data)
        rows += [n * n2 + n] * m; cols += list(range(1, m + 1)); vals += list(-trA)
        rows += list(np.arange(n) * n2 + np.arange(n)) + [n2 * n2 - 1]; cols += [m + 1] * (n + 1); vals += [1.0] * (n + 1)
        P1 = SDP()
        P1._A = sp.csc_matrix((vals, (rows, cols)), shape=(n2 * n2, m + 2))
        P1._b = np.concatenate([b - k * trA, [MM]])
        P1._blockstruct = [n, -2]
        # strictly feasible start of the Phase-I problem: X0 + t I completable (bisection on t), then lift
        tmin, tmax = 0.0, 1.0
        eye = sp.identity(n, format="csc")
        while True:
            t = 0.5 * (tmin + tmax)
            if feasible(X0 + t * eye):
                tmax = t
                if tmax - tmin < 1e-1:
                    break
            else:
                tmax *= 2.0
                tmin = t
        tt = t + 1.0
        U = sym(X0) + tt * eye
        trU = U.diagonal().sum()
        Z0 = sp.block_diag([U, sp.diags([tt + k, MM - trU])], format="csc")
        sol = P1.solve_feas(primalstart={"x": Z0}, kktsolver=kktsolver)
        Xs = sp.csc_matrix(sol["x"])
        s_ = Xs[n, n] - k
        if s_ > 0:
            return None, P1
        sol.pop("y", None)
        sol.pop("s", None)
        X = sp.csc_matrix(sol.pop("x"))[:n, :n] - s_ * eye
        return sp.csc_matrix(X), sol

    def write_sdpa(self, fname=None, compress=False):
        """Writes the problem to fname + '.dat-s' (base.py:196-217): negated data (neg=True, the inverse of what
        SDP(filename) applies), refuses to overwrite, optional bz2 compression."""
        if not self._blockstruct:
            self._blockstruct = [self.n]
        if fname is None:
            fname = self._pname
        fname = str(fname) + ".dat-s"
        if os.path.isfile(fname):
            raise IOError("file %s already exists" % fname)
        if compress and os.path.isfile(fname + ".bz2"):
            raise IOError("file %s already exists" % (fname + ".bz2"))
        sdpa_write(fname, self._A, self._b, self._blockstruct, neg=True)
        if compress:
            import bz2
            with open(fname, "rb") as fi, open(fname + ".bz2", "wb") as fo:
                fo.write(bz2.compress(fi.read()))
            os.remove(fname)


def _band_entries(n, bw):
    J = np.repeat(np.arange(n), np.minimum(bw + 1, n - np.arange(n)))
    I = np.concatenate([np.arange(j, min(j + bw + 1, n)) for j in range(n)])
    return I.astype(np.int64), J.astype(np.int64)


def _band_posdef(n, bw, rng):
    """Random positive definite band matrix (dense storage): diagonally dominant."""
    M = np.zeros((n, n))
    for d in range(1, bw + 1):
        v = rng.standard_normal(n - d) / math.sqrt(n)
        M[np.arange(d, n), np.arange(n - d)] = v
    M = M + M.T
    M[np.diag_indices(n)] = np.abs(M).sum(axis=1) + 0.1 + rng.random(n)
    return M


class band_SDP(SDP):
    """Random SDP with band structure: P = band_SDP(n, m, bw, seed) (bw = half bandwidth).
    Same construction as the reference (base.py:600-636): strictly feasible X0 / (y0, S0) by
    construction, A_i dense on the band with std 1/|V|, C = sum_i y0_i A_i + S0, b = A(X0)."""

    def __init__(self, n, m, bw, seed=0):
        super().__init__()
        rng = np.random.default_rng(seed)
        I, J = _band_entries(n, bw)
        nv = len(I)
        y0 = rng.standard_normal(m)
        y0 /= np.linalg.norm(y0)
        S0 = _band_posdef(n, bw, rng)
        X0 = _band_posdef(n, bw, rng)        # a positive definite band matrix is trivially completable
        Av = rng.standard_normal((nv, m)) / nv
        c = S0[I, J] + Av @ y0
        w = np.where(I == J, 1.0, 2.0)
        b = Av.T @ (w * X0[I, J])
        rows = I + n * J
        data = np.concatenate([c[:, None], Av], axis=1)
        self._A = sp.csc_matrix((data.reshape(-1, order="F"),
                                 (np.tile(rows, m + 1), np.repeat(np.arange(m + 1), nv))),
                                shape=(n * n, m + 1))
        self._b = b
        self._bw = bw
        self._blockstruct = [n]
        self._pname = "band_n%i_m%i_bw%i" % (n, m, bw)
        self._X0, self._y0, self._S0 = X0, y0, S0

    @property
    def bw(self):
        return self._bw


class mtxnorm_SDP(SDP):
    """Matrix norm minimization  minimize || A_1 y_1 + ... + A_r y_r + B ||_2  (p x q matrices) as the SDP

        minimize t   subject to   [ t I  (A(y)+B)' ; A(y)+B  t I ] >= 0      (order n = p + q)

    P = mtxnorm_SDP(p, q, r, density=1.0, seed=0).  Same construction as the reference (base.py:639-773): column 0
    of A holds B in the (2,1) block (rows q..n-1, columns 0..q-1), columns 1..r the data matrices with
    nz = min(max(1, round(density p q)), p q) random entries of that block each, column r+1 is -I, b = (0,...,0,-1);
    values are standard normal (seeded numpy generator -- the reference's cvxopt generator is not reproducible here).
    Structural known answer of the reference's documentation (docs.rst:596,608): mtxnorm_SDP(200, 10, 200) has
    n = 210, m = 201, nnz = 2210."""

    def __init__(self, p, q, r, density=1.0, seed=0):
        super().__init__()
        if isinstance(density, float):
            if density > 1 or density <= 0:
                raise ValueError("density must be between 0 and 1")
            dens = [density] * r
            self._pname = ("mtxnorm_p%i_q%i_r%i" % (p, q, r) if density == 1.0
                           else "mtxnorm_p%i_q%i_r%i_d%i" % (p, q, r, int(density * 1000)))
        elif isinstance(density, list):
            if len(density) != r:
                raise TypeError("density must be a float between 0 and 1 or a list of r floats")
            dens = density
            self._pname = "mtxnorm_p%i_q%i_r%i_vd" % (p, q, r)
        else:
            raise TypeError("density must be a float between 0 and 1 or a list of r floats")
        if not isinstance(seed, int):
            raise ValueError("seed must be an integer")
        rng = np.random.default_rng(seed)
        n = p + q
        I1 = np.tile(np.arange(q, n), q)                 # (2,1) block, column-major: rows q..n-1 of columns 0..q-1
        J1 = np.repeat(np.arange(q), p)
        lin = I1 + n * J1
        rows, cols, vals = [lin], [np.zeros(p * q, dtype=np.int64)], [rng.standard_normal(p * q)]
        for j in range(r):
            nz = min(max(1, int(round(dens[j] * p * q))), p * q)
            rows.append(rng.choice(lin, size=nz, replace=False))
            cols.append(np.full(nz, j + 1, dtype=np.int64))
            vals.append(rng.standard_normal(nz))
        rows.append(np.arange(0, n * n, n + 1))
        cols.append(np.full(n, r + 1, dtype=np.int64))
        vals.append(-np.ones(n))
        self._A = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n * n, r + 2))
        b = np.zeros(r + 1)
        b[-1] = -1.0
        self._b = b
        self._p, self._q, self._density = p, q, density
        self._blockstruct = [n]


def completion(X):
    """Maximum-determinant positive definite completion of the sparse symmetric matrix X (scipy sparse, either
    triangle or both), returned as a dense numpy array; ArithmeticError if X has no positive definite completion.
    Counterpart of smcp.completion (base.py:952-973): embed the pattern (perfect elimination order if it is chordal,
    minimum degree otherwise), chompack-level completion -> factor L of the inverse, then Z = (L L^T)^-1 by two
    supernodal triangular solves with the identity."""
    import torch
    from . import chordal
    from .cspmatrix import cspmatrix
    from .symbolic import Symbolic, maxcardsearch, mindegree
    X = sp.csc_matrix(X)
    n = X.shape[0]
    Xl = sp.tril(X + sp.triu(X, 1).T if (sp.triu(X, 1).nnz and not sp.tril(X, -1).nnz) else X).tocoo()
    pat = sp.csc_matrix((np.ones(Xl.nnz), (Xl.row, Xl.col)), shape=(n, n)) + sp.identity(n, format="csc")
    symb = Symbolic(pat, maxcardsearch(pat))
    if symb.fill > 0:
        symb = Symbolic(pat, mindegree(pat))
    dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    L = cspmatrix.from_entries(symb, Xl.row, Xl.col, Xl.data, device=dev)
    chordal.completion(L)
    B = torch.eye(n, dtype=torch.float64, device=dev)          # rows = right-hand sides, PERMUTED coordinates
    chordal.trsm(L, B)
    chordal.trsm(L, B, trans="T")
    Z = B.cpu().numpy()
    ip = np.asarray(symb.ip)
    Z = Z[np.ix_(ip, ip)]
    return 0.5 * (Z + Z.T)


class pattern_SDP(SDP):
    """Random strictly feasible SDP on a given chordal sparsity pattern: P = pattern_SDP(pat, m, density, seed) with
    pat = (n, colptr, rowind) (lower triangle).  The band_SDP recipe (base.py:600-636) on any pattern: X0 and S0 are
    positive definite with pattern V (L L^T of a random factor on V, formed on the device), the A_i have
    max(1, round(density |V|)) random entries of V each (the reference's UFSMC recipe, doc benchmarks index.rst:479),
    b = A(X0), C = S0 + sum_i y0_i A_i.  Used for whole interior-point runs on the benchmark patterns (synth50k)."""

    def __init__(self, pat, m, density=0.005, seed=0):
        super().__init__()
        import torch
        from . import chordal, problems
        from .cspmatrix import cspmatrix
        from .symbolic import Symbolic
        n, cp, ri = pat
        rng = np.random.default_rng(seed)
        J = np.repeat(np.arange(n, dtype=np.int64), np.diff(cp))
        I = np.asarray(ri, dtype=np.int64)
        nv = len(I)
        symb = Symbolic(pat)
        if symb.fill:
            raise ValueError("pattern_SDP needs a chordal pattern in a perfect elimination order")

        def posdef(sd):
            # (without a GPU the tensor stays on the host and chordal.llt fails loudly -- unless the tests' oracle backend serves it)
            X = cspmatrix(symb, torch.from_numpy(problems.random_factor_blkval(symb, sd)).to("cuda" if torch.cuda.is_available() else "cpu"))
            chordal.llt(X)
            return sp.csc_matrix(X.spmatrix(reordered=False, symmetric=False))

        X0, S0 = posdef(seed + 1), posdef(seed + 2)
        per = max(1, int(round(density * nv)))
        y0 = rng.standard_normal(m)
        y0 /= np.linalg.norm(y0)
        x0v = np.asarray(X0[I, J]).ravel()
        rows, cols, vals = [], [], []
        b = np.zeros(m)
        csum = np.zeros(nv)
        for i in range(m):
            sel = np.sort(rng.choice(nv, size=per, replace=False))
            v = rng.standard_normal(per) / np.sqrt(per)
            rows.append(I[sel] + n * J[sel])
            cols.append(np.full(per, i + 1, dtype=np.int64))
            vals.append(v)
            b[i] = np.sum(np.where(I[sel] == J[sel], 1.0, 2.0) * v * x0v[sel])
            csum[sel] += y0[i] * v
        rows.insert(0, I + n * J)
        cols.insert(0, np.zeros(nv, dtype=np.int64))
        vals.insert(0, np.asarray(S0[I, J]).ravel() + csum)
        self._A = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n * n, m + 1))
        self._b = b
        self._blockstruct = [n]
        self._pname = "pattern_n%i_m%i" % (n, m)
        self._X0, self._y0, self._S0 = X0, y0, S0


def maxcut_SDP(n=1000, nedges=5909, seed=0):
    """Max-cut relaxation on a random graph with the size of SDPLIB maxG51 (config 4):
    minimize <C,X>, diag(X) = 1, C = -(Diag(W1) - W)/4."""
    from .problems import maxcut_graph_pattern
    _, e = maxcut_graph_pattern(n, nedges, seed)
    P = SDP()
    W = sp.coo_matrix((np.ones(len(e)), (e[:, 0], e[:, 1])), shape=(n, n))
    W = W + W.T
    deg = np.asarray(W.sum(axis=1)).reshape(-1)
    Cm = sp.tril(-(sp.diags(deg) - W) / 4.0).tocoo()
    rows = [Cm.row + n * Cm.col]
    cols = [np.zeros(len(Cm.row), dtype=np.int64)]
    vals = [Cm.data]
    rows.append(np.arange(n) * (n + 1))
    cols.append(np.arange(1, n + 1))
    vals.append(np.ones(n))
    P._A = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n * n, n + 1))
    P._b = np.ones(n)
    P._blockstruct = [n]
    P._pname = "maxcut_n%d_e%d" % (n, nedges)
    return P


# ---- SDPA sparse format (misc.c:56-365) -----------------------------------------------------
# A number as C's scanf("%d") / scanf("%lf") accepts it after the reference's skip of everything that is not a
# digit or a sign (the "%*[^0-9+-]" directive at misc.c:94,176,185,205-209): SDPLIB headers such as "2 =mdim",
# "{2, -2}" or "(1.0, 2.0)" parse because the text between the numbers is skipped, not tokenised.
_SDPA_NUM = re.compile(r"[-+]?(?:\d+\.?\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?)")


def _sdpa_numbers(filename):
    """(m, iterator over the numbers after the line that holds m).  Leading comment lines start with '*' or '"'
    (misc.c:76-82, 158-164); m is the first integer of the first other line, the rest of that line is dropped."""
    with open(filename) as f:
        text = f.read()
    pos = 0
    m = None
    while pos < len(text):
        end = text.find("\n", pos)
        end = len(text) if end < 0 else end
        line = text[pos:end]
        pos = end + 1
        if line[:1] in ('*', '"'):
            continue
        mt = re.match(r"\s*([-+]?\d+)", line)
        if mt is None:
            raise ValueError("SDPA file %s: expected the number of constraints, got %r" % (filename, line[:40]))
        m = int(mt.group(1))
        break
    if m is None:
        raise ValueError("SDPA file %s: no header" % filename)
    return m, (mt.group(0) for mt in _SDPA_NUM.finditer(text, pos))


def _sdpa_int(tok):
    return int(float(tok))


def sdpa_readhead(filename):
    """(n, m, blockstruct) from the header (misc.c:56-103)."""
    m, toks = _sdpa_numbers(filename)
    nb = _sdpa_int(next(toks))
    bs = [_sdpa_int(next(toks)) for _ in range(nb)]
    return sum(abs(x) for x in bs), m, bs


def sdpa_read(filename, neg=False):
    """Returns (A, b, blockstruct): A is n^2 x (m+1) CSC with columns vec(lower triangles); the SDPA entry
    ``matno blkno i j val`` (1-based, upper triangle of block blkno) lands at row (off+j-1) + n (off+i-1), the
    lower-triangular position of the symmetric entry (misc.c:205-237; an entry given in the lower triangle is
    mirrored instead of being stored above the diagonal).  A negative block size is a diagonal block of that
    many rows (misc.c:177-178).  Explicit zeros are dropped (misc.c:216).  neg=True negates b and A
    (misc.c:186-187, 223-224) -- what SDP(filename) asks for, base.py:189,194.  A truncated last record ends
    the data as the reference's break at misc.c:205-209 does."""
    m, toks = _sdpa_numbers(filename)
    try:
        nb = _sdpa_int(next(toks))
        bs = [_sdpa_int(next(toks)) for _ in range(nb)]
        offs = np.concatenate([[0], np.cumsum([abs(x) for x in bs])]).astype(np.int64)
        n = int(offs[-1])
        b = np.array([float(next(toks)) for _ in range(m)])
    except StopIteration:
        raise ValueError("SDPA file %s: truncated header" % filename)
    rest = list(toks)
    nrec = len(rest) // 5
    rec = np.array([float(t) for t in rest[:5 * nrec]]).reshape(nrec, 5)
    mat, blk = rec[:, 0].astype(np.int64), rec[:, 1].astype(np.int64)
    keep = rec[:, 4] != 0.0
    if nrec and (mat.min() < 0 or mat.max() > m or blk.min() < 1 or blk.max() > len(bs)):
        raise ValueError("SDPA file %s: matrix or block number out of range" % filename)
    a = offs[blk - 1] + rec[:, 2].astype(np.int64) - 1
    c = offs[blk - 1] + rec[:, 3].astype(np.int64) - 1
    if nrec and (np.minimum(a, c).min() < 0 or (np.maximum(a, c) >= offs[blk]).any()):
        raise ValueError("SDPA file %s: entry outside its block" % filename)
    lo, hi = np.minimum(a, c)[keep], np.maximum(a, c)[keep]
    sgn = -1.0 if neg else 1.0
    A = sp.csc_matrix((sgn * rec[keep, 4], (hi + n * lo, mat[keep])), shape=(n * n, m + 1))
    return A, sgn * b, bs


def sdpa_write(filename, A, b, blockstruct, neg=False):
    """Write the problem in SDPA sparse format with the reference's header (misc.c:303-323; the reference opens
    the file with mode "r", misc.c:299 -- a bug this writer does not share).  Zero entries are skipped
    (misc.c:350); values with 17 significant digits so that a write/read round trip is exact."""
    A = sp.csc_matrix(A)
    n = int(round(math.sqrt(A.shape[0])))
    m = A.shape[1] - 1
    offs = np.concatenate([[0], np.cumsum([abs(x) for x in blockstruct])])
    sgn = -1.0 if neg else 1.0
    with open(filename, "w") as f:
        f.write("* sparse SDPA data file (created by smcp_amd)\n")
        f.write("%d = m\n%d = nBlocks\n%s\n" % (m, len(blockstruct), " ".join(str(int(x)) for x in blockstruct)))
        f.write(" ".join("%.17g" % float(sgn * v) for v in np.asarray(b).reshape(-1)) + "\n")
        for k in range(m + 1):
            col = A[:, k].tocoo()
            order = np.argsort(col.row, kind="stable")
            for r, v in zip(col.row[order], col.data[order]):
                if v == 0.0:
                    continue
                i, j = r % n, r // n            # i >= j (lower); SDPA wants upper: (j, i)
                if j > i:
                    i, j = j, i
                blk = int(np.searchsorted(offs, j, side="right"))
                if i >= offs[blk]:
                    raise ValueError("sdpa_write: entry (%d, %d) lies outside the diagonal blocks" % (i, j))
                f.write("%d %d %d %d %.17g\n" % (k, blk, j - offs[blk - 1] + 1, i - offs[blk - 1] + 1, float(sgn * v)))

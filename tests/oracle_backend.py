"""Test-only backend: routes the CHOMPACK-named entry points of smcp_amd.chordal / smcp_amd.kkt to the
CPU oracle so that the HOST logic (interior-point drivers, index algebra, sharding plans) can be
exercised by the ``-m "not gpu"`` suite.  Never imported by the product."""
import contextlib

import numpy as np
import torch

from oracle import oracle as orc
from smcp_amd import chordal, kkt, solvers
from smcp_amd.cspmatrix import cspmatrix

def _S(symb):
    if "_orc_sym" not in symb.__dict__:
        symb.__dict__["_orc_sym"] = orc.Sym(symb)
    return symb.__dict__["_orc_sym"]


def _np(X):
    return X.blkval.numpy()


class OracleKKT(kkt.ShardedSchur):
    """Same host-side sharding logic as the product's KKTSystem, compute by the CPU oracle."""

    def __init__(self, symb, cptr, cidx, cval, max_rhs=None, tnzcols=None):
        self.symb = symb
        self.m = len(cptr) - 1
        self.K = orc.KKT(_S(symb), np.asarray(cptr), np.asarray(cidx), np.asarray(cval))
        self.dev = torch.device("cpu")
        self._Hbuf = torch.zeros(self.m * self.m + 1, dtype=torch.float64)     # + the status word of the all-reduce
        self.H = self._Hbuf[:self.m * self.m].view(self.m, self.m)             # H.T is the column-major matrix

    def amap(self, X):
        return torch.from_numpy(self.K.amap(_np(X)))

    def aadj(self, y):
        return cspmatrix(self.symb, torch.from_numpy(self.K.aadj(y.numpy())))

    def _columns(self, L, Y, j0, j1):
        for j in range(j0, j1):
            u = self.K.constraint(j)
            orc.hessian(self.K.S, _np(L), _np(Y), u, adj=None, inv=False)
            self.H[j, :] = torch.from_numpy(self.K.amap(u))

    # ---- column-sparse constraints sharded by constraint (kkt_schur_gram_part): emulated with every constraint treated
    # as column-sparse when emulate_sparse is set -- the columns come from the oracle's Hessian, the ownership rule (a pair
    # owned by two ranks is written by the owner of the smaller index) is the product's (k_scm_columns)
    emulate_sparse = False

    def _sparse_count(self):
        return self.m if self.emulate_sparse else 0

    def _scm_part(self, L, Y, part, nparts):
        m = self.m
        owner = np.empty(m, dtype=int)
        for p in range(nparts):
            owner[m * p // nparts: m * (p + 1) // nparts] = p
        for s_ in range(m * part // nparts, m * (part + 1) // nparts):
            u = self.K.constraint(s_)
            orc.hessian(self.K.S, _np(L), _np(Y), u, adj=None, inv=False)
            col = self.K.amap(u)
            for i in range(m):
                if owner[i] != part and i < s_:
                    continue
                self.H[s_, i] = col[i]
                self.H[i, s_] = col[i]

    # ---- subtree-sharded Gram path, emulated with the oracle's masked G sweep
    def _apply_partition(self, P, rank):
        self._mask = {1: (P.owner == rank).astype(np.uint8), 2: (P.owner == -1).astype(np.uint8)}

    def _gram_chunk(self):
        return 3                              # forces several chunks in the tests

    def _gram_prepare(self, L, Y):
        S = self.K.S
        self._L = _np(L)
        self._yaa, self._fac = orc.prepare_fac(S, _np(Y))
        self._G = [self.K.constraint(j) for j in range(self.m)]
        self._upd = [np.zeros(max(1, S.updlen)) for _ in range(self.m)]

    def _gram_sweep(self, which, j0, j1):
        for j in range(j0, j1):
            orc.hess_g_masked(self.K.S, self._L, self._fac, self._G[j], self._upd[j], self._mask[which])
        self._xbufs = self._upd[j0:j1]          # what the next boundary exchange carries

    def _exchange_pack(self, cliques, nrhs, out):
        S = self.K.S
        assert nrhs == len(self._xbufs)
        parts = [u[S.updptr[k]:S.updptr[k + 1]] for k in cliques for u in self._xbufs]
        out.copy_(torch.from_numpy(np.concatenate(parts)))

    def _exchange_size(self, cliques, nrhs):        # the oracle exchanges full squares
        S = self.K.S
        return int(sum(int(S.updptr[k + 1] - S.updptr[k]) for k in cliques) * nrhs)

    def _exchange_unpack(self, cliques, nrhs, buf):
        S = self.K.S
        b, o = buf.numpy(), 0
        for k in cliques:
            n = int(S.updptr[k + 1] - S.updptr[k])
            for u in self._xbufs:
                u[S.updptr[k]:S.updptr[k + 1]] = b[o:o + n]
                o += n

    def _exchange_pack_range(self, cliques, r0, nrhs, out):
        S = self.K.S
        parts = [u[S.updptr[k]:S.updptr[k + 1]] for k in cliques for u in self._xbufs[r0:r0 + nrhs]]
        out.copy_(torch.from_numpy(np.concatenate(parts)))

    def _exchange_unpack_share(self, P, lo, hi, recv, width):
        """numpy form of csp_exchange_unpack_all: the roots of EVERY rank, into the update arrays of the constraints lo .. hi - 1"""
        S = self.K.S
        b = recv.numpy()
        for r, cliques in enumerate(P.roots_by_rank):
            o = r * width
            for k in cliques:
                n = int(S.updptr[k + 1] - S.updptr[k])
                for u in self._upd[lo:hi]:
                    u[S.updptr[k]:S.updptr[k + 1]] = b[o:o + n]
                    o += n

    def _stack_rows(self, direction, j0, j1, a, b, buf):
        v = buf.numpy()
        for j in range(j0, j1):
            seg = slice((j - j0) * (b - a), (j - j0 + 1) * (b - a))
            if direction:
                self._G[j][a:b] = v[seg]
            else:
                v[seg] = self._G[j][a:b]

    def _exchange_combine(self, P, rank, nrhs, y, recv, width, out, owidth, mode):
        """numpy form of csp_exchange_combine on the oracle's exchange layout ([clique][rhs][square block])"""
        S = self.K.S
        g, o, yv = recv.numpy(), out.numpy(), y.numpy()
        for r, cliques in enumerate(P.roots_by_rank):
            if r == rank:
                continue
            og, oo = r * width, r * owidth
            for k in cliques:
                n = int(S.updptr[k + 1] - S.updptr[k])
                acc = yv @ g[og:og + nrhs * n].reshape(nrhs, n)
                o[oo:oo + n] = (o[oo:oo + n] + acc) if mode else (acc - o[oo:oo + n])
                og += nrhs * n
                oo += n

    # ---- sharded factorisation / solve, emulated with the oracle's masked sweeps
    def _upd1(self):
        if "_u1" not in self.__dict__:
            self._u1 = np.zeros(max(1, self.K.S.updlen))
        self._xbufs = [self._u1]
        return self._u1

    def _chol_part(self, L, which):
        orc.cholesky_masked(self.K.S, _np(L), self._upd1(), self._mask[which])

    def _pinv_part(self, Y, which):
        orc.projected_inverse_masked(self.K.S, _np(Y), self._upd1(), self._mask[which])

    def _prepare_part(self, L, Y, which):
        S = self.K.S
        if which == 2:
            self._L = _np(L)
            self._yaa, self._fac = np.zeros(max(1, S.updlen)), np.zeros(max(1, S.updlen))
        orc.prepare_fac_masked(S, _np(Y), self._yaa, self._fac, self._mask[which])

    def _gram_prepare_part(self):
        S = self.K.S
        self._G = [self.K.constraint(j) for j in range(self.m)]
        self._upd = [np.zeros(max(1, S.updlen)) for _ in range(self.m)]

    def _hess_part(self, U, which, direction):
        if direction == 0:
            orc.hess_up_masked(self.K.S, self._L, self._yaa, _np(U), self._upd1(), self._mask[which])
        else:
            orc.hess_down_masked(self.K.S, self._L, _np(U), self._upd1(), self._mask[which])

    def _potrs(self, y):
        H = np.asfortranarray(self.H.numpy().T)
        b = np.asfortranarray(y.numpy().copy())
        orc.dense_potrs(H, b)
        y.copy_(torch.from_numpy(b))

    def _own(self):
        pass

    def _gram_accumulate(self, ranges):
        S = self.K.S
        w = np.zeros(S.blklen)
        for k in range(S.nsn):
            nn = int(S.snptr[k + 1] - S.snptr[k])
            nf = int(S.rowptr[k + 1] - S.rowptr[k])
            blk = np.full((nf, nn), 2.0)
            blk[:nn, :nn] = np.tril(np.full((nn, nn), 2.0), -1) + np.eye(nn)
            w[S.blkptr[k]:S.blkptr[k] + nf * nn] = blk.reshape(-1, order="F")
        G = np.stack(self._G, axis=1)                       # blklen x m
        H = np.zeros((self.m, self.m))
        for a, b in ranges:
            H += G[a:b].T @ (w[a:b, None] * G[a:b])
        self.H.copy_(torch.from_numpy(H))

    def _potrf(self):
        Hf = np.asfortranarray(self.H.numpy().T)
        orc.dense_potrf(Hf)
        self.H.copy_(torch.from_numpy(np.ascontiguousarray(Hf.T)))

    def factor(self, L, Y, group=None):
        self.build_schur(L, Y, group)
        self._potrf()
        if self._sharded_pair(L, Y):
            kept, gen = getattr(self, "_kept", None), self.__dict__.get("_kept_gen", 0)
            return lambda bx, by, kk, complete=True: self._solve_sharded(
                L, Y, bx, by, kk, group, complete, kept if self.__dict__.get("_kept_gen", 0) == gen else None)
        H = np.asfortranarray(self.H.numpy().T)

        def solve_(bx, by, kk):
            x, y = self.K.solve(_np(L), _np(Y), H, _np(bx).copy(), by.numpy().copy(), kk)
            bx.blkval.copy_(torch.from_numpy(x))
            by.copy_(torch.from_numpy(y))
            return bx, by

        return solve_

    def factor_qr(self, L, Y, group=None):
        F = self.K.qr_factor(_np(L), _np(Y))

        def solve_(bx, by, kk):
            x, y = self.K.qr_solve(_np(L), _np(Y), F, _np(bx).copy(), by.numpy().copy(), kk)
            bx.blkval.copy_(torch.from_numpy(x))
            by.copy_(torch.from_numpy(y))
            return bx, by

        return solve_


@contextlib.contextmanager
def oracle_backend():
    saved = {k: getattr(chordal, k) for k in ("cholesky", "llt", "projected_inverse", "completion", "hessian",
                                              "dot", "logdiagsum", "trsm", "cholesky_projected_inverse")}
    saved_kkt, saved_skkt = kkt.KKTSystem, solvers.KKTSystem
    chordal.cholesky = lambda X: orc.cholesky(_S(X.symb), _np(X))
    chordal.llt = lambda X: orc.llt(_S(X.symb), _np(X))
    chordal.projected_inverse = lambda X: orc.projected_inverse(_S(X.symb), _np(X))
    chordal.completion = lambda X: orc.completion(_S(X.symb), _np(X))

    def cholesky_projected_inverse(L, Y, factors=True):
        orc.cholesky(_S(L.symb), _np(L))
        Y.blkval.copy_(L.blkval)
        orc.projected_inverse(_S(Y.symb), _np(Y))

    chordal.cholesky_projected_inverse = cholesky_projected_inverse

    def hessian(L, Y, U, adj=False, inv=False):
        Us = [U] if isinstance(U, cspmatrix) else U
        for u in Us:
            orc.hessian(_S(L.symb), _np(L), _np(Y), _np(u), adj=adj, inv=inv)

    chordal.hessian = hessian
    chordal.trsm = lambda L, B, trans="N": orc.trsm(_S(L.symb), _np(L), B.numpy(), trans)

    def probe_factors(base, d, alphas, kind):       # the trial factorisations of one probe, emulated one after the other
        ok, fac = [], []
        for al in alphas:
            t = _np(base) + al * _np(d)
            try:
                (orc.completion if kind == "p" else orc.cholesky)(_S(base.symb), t)
                ok.append(True)
                fac.append(cspmatrix(base.symb, torch.from_numpy(t)))
            except ArithmeticError:
                ok.append(False)
                fac.append(None)
        return ok, fac

    def probe_cone(base, d, alphas, kind):
        return probe_factors(base, d, alphas, kind)[0]

    saved["probe_factors"] = getattr(chordal, "probe_factors", None)
    chordal.probe_factors = probe_factors
    saved["probe_cone"] = getattr(chordal, "probe_cone", None)
    saved["_probe_emulated"] = getattr(chordal, "_probe_emulated", False)
    chordal.probe_cone = probe_cone
    chordal._probe_emulated = True
    chordal.dot = lambda X, Y: orc.dot(_S(X.symb), _np(X), _np(Y))
    chordal.logdiagsum = lambda X: orc.logdiagsum(_S(X.symb), _np(X))
    kkt.KKTSystem = OracleKKT
    solvers.KKTSystem = OracleKKT
    try:
        yield
    finally:
        for k, v in saved.items():
            setattr(chordal, k, v)
        kkt.KKTSystem, solvers.KKTSystem = saved_kkt, saved_skkt

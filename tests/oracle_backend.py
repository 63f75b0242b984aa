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

    def __init__(self, symb, cptr, cidx, cval, max_rhs=None):
        self.symb = symb
        self.m = len(cptr) - 1
        self.K = orc.KKT(_S(symb), np.asarray(cptr), np.asarray(cidx), np.asarray(cval))
        self.dev = torch.device("cpu")
        self.H = torch.zeros((self.m, self.m), dtype=torch.float64)   # H.T is the column-major matrix

    def amap(self, X):
        return torch.from_numpy(self.K.amap(_np(X)))

    def aadj(self, y):
        return cspmatrix(self.symb, torch.from_numpy(self.K.aadj(y.numpy())))

    def _columns(self, L, Y, j0, j1):
        for j in range(j0, j1):
            u = self.K.constraint(j)
            orc.hessian(self.K.S, _np(L), _np(Y), u, adj=None, inv=False)
            self.H[j, :] = torch.from_numpy(self.K.amap(u))

    def _potrf(self):
        Hf = np.asfortranarray(self.H.numpy().T)
        orc.dense_potrf(Hf)
        self.H.copy_(torch.from_numpy(np.ascontiguousarray(Hf.T)))

    def factor(self, L, Y, group=None):
        self.build_schur(L, Y, group)
        self._potrf()
        H = np.asfortranarray(self.H.numpy().T)

        def solve_(bx, by, kk):
            x, y = self.K.solve(_np(L), _np(Y), H, _np(bx).copy(), by.numpy().copy(), kk)
            bx.blkval.copy_(torch.from_numpy(x))
            by.copy_(torch.from_numpy(y))
            return bx, by

        return solve_


@contextlib.contextmanager
def oracle_backend():
    saved = {k: getattr(chordal, k) for k in ("cholesky", "llt", "projected_inverse", "completion", "hessian",
                                              "dot", "logdiagsum", "trsm")}
    saved_kkt, saved_skkt = kkt.KKTSystem, solvers.KKTSystem
    chordal.cholesky = lambda X: orc.cholesky(_S(X.symb), _np(X))
    chordal.llt = lambda X: orc.llt(_S(X.symb), _np(X))
    chordal.projected_inverse = lambda X: orc.projected_inverse(_S(X.symb), _np(X))
    chordal.completion = lambda X: orc.completion(_S(X.symb), _np(X))

    def hessian(L, Y, U, adj=False, inv=False):
        Us = [U] if isinstance(U, cspmatrix) else U
        for u in Us:
            orc.hessian(_S(L.symb), _np(L), _np(Y), _np(u), adj=adj, inv=inv)

    chordal.hessian = hessian
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

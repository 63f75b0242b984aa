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


class OracleKKT:
    def __init__(self, symb, cptr, cidx, cval, max_rhs=None):
        self.symb = symb
        self.m = len(cptr) - 1
        self.K = orc.KKT(_S(symb), np.asarray(cptr), np.asarray(cidx), np.asarray(cval))
        self.dev = torch.device("cpu")
        self.H = None

    def amap(self, X):
        return torch.from_numpy(self.K.amap(_np(X)))

    def aadj(self, y):
        return cspmatrix(self.symb, torch.from_numpy(self.K.aadj(y.numpy())))

    def factor(self, L, Y):
        H = self.K.schur_factor(_np(L), _np(Y))
        self.H = H

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

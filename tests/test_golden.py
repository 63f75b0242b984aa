"""Oracle and HIP path against the committed fixtures tests/golden/*.npz.

The fixtures are dense float64 restatements of the DEFINITIONS of the calls on the Newton-KKT path (generator:
tests/golden/make_golden.py; neither the oracle nor the HIP library takes part in producing them).  They are not
outputs of the reference -- CHOMPACK / CVXOPT cannot be installed here and the reference ships no golden vectors for
this path -- so they pin the oracle and the kernels to the mathematics, not to the reference's bits.
Tolerance: 1e-10 relative (fp64, condition numbers <= 3e2, printed by the generator)."""
import glob
import os

import numpy as np
import pytest

from oracle import oracle as orc
from smcp_amd.symbolic import Symbolic

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))
TOL = 1e-10


def rel(a, b):
    return np.linalg.norm(a - b) / max(1.0, np.linalg.norm(b))


class Case:
    """Fixture with its dense matrices moved to the PERMUTED coordinates of the symbolic analysis (which
    postorders the supernodes): M_perm = M[p][:, p].  chol is recomputed for that order (the Cholesky factor is
    not permutation covariant) and checked against the stored one where p is the identity."""

    def __init__(self, path):
        g = np.load(path)
        self.symb = Symbolic((int(g["n"]), g["colptr"].astype(np.int64), g["rowind"].astype(np.int64)))
        assert self.symb.fill == 0             # the fixtures' natural order is a perfect elimination order
        p = np.asarray(self.symb.p)
        P = lambda M: M[np.ix_(p, p)]
        self.S, self.projinv, self.U, self.hessU = P(g["S"]), P(g["projinv"]), P(g["U"]), P(g["hessU"])
        self.bx, self.x, self.hinv_x = P(g["bx"]), P(g["x"]), P(g["hinv_x"])
        self.A = g["A"]                        # constraints stay in ORIGINAL coordinates (index_map takes those)
        self.schur, self.by, self.y, self.kk = g["schur"], g["by"], g["y"], g["kk"]
        self.chol = np.linalg.cholesky(self.S)
        if (p == np.arange(len(p))).all():
            assert rel(self.chol, g["chol"]) < 1e-13

    def __getitem__(self, k):
        return getattr(self, k)


def load(path):
    c = Case(path)
    return c, c.symb


def constraints(symb, A):
    """A (m, n, n) dense symmetric -> CSC over blkval positions (lower entries, values as given)."""
    cptr, cidx, cval = [0], [], []
    for Ai in A:
        I, J = np.nonzero(np.tril(Ai))
        cidx.append(symb.index_map(I, J))
        cval.append(Ai[I, J])
        cptr.append(cptr[-1] + len(I))
    return np.asarray(cptr, dtype=np.int64), np.concatenate(cidx).astype(np.int64), np.concatenate(cval)


def test_fixtures_present():
    assert len(FILES) >= 4


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_oracle_against_golden(path):
    g, symb = load(path)
    S = orc.Sym(symb)
    x = S.project(g["S"])
    orc.cholesky(S, x)
    assert rel(S.dense(x, symmetric=False), g["chol"]) < TOL
    L = x.copy()
    orc.projected_inverse(S, x)
    assert rel(S.dense(x), g["projinv"]) < TOL
    Y = x.copy()
    orc.completion(S, x)                                   # X = P_V(S^-1): the completion's inverse is S
    assert rel(S.dense(x, symmetric=False), g["chol"]) < 1e-9
    u = S.project(g["U"])
    orc.hessian(S, L, Y, u, adj=None, inv=False)
    assert rel(S.dense(u), g["hessU"]) < TOL
    orc.hessian(S, L, Y, u, adj=None, inv=True)
    assert rel(S.dense(u), g["U"]) < 1e-9
    cptr, cidx, cval = constraints(symb, g["A"])
    K = orc.KKT(S, cptr, cidx, cval)
    Hf = K.schur_factor(L, Y)
    assert rel(np.tril(Hf), np.linalg.cholesky(g["schur"])) < TOL
    xs, ys = K.solve(L, Y, Hf, S.project(g["bx"]), g["by"].copy(), g["kk"])
    assert rel(S.dense(xs), g["x"]) < 1e-9 and rel(ys, g["y"]) < 1e-9
    hx = xs.copy()
    orc.hessian(S, L, Y, hx, adj=None, inv=True)
    assert rel(S.dense(hx), g["hinv_x"]) < 1e-8


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_hip_against_golden(path):
    import torch
    from smcp_amd import chordal
    from smcp_amd.cspmatrix import cspmatrix
    from smcp_amd.kkt import KKTSystem
    g, symb = load(path)
    symb.device_init(0, 4)
    S = orc.Sym(symb)                                      # layout helper only (project / dense)
    dev = lambda v: cspmatrix(symb, torch.from_numpy(np.ascontiguousarray(v)).cuda())
    host = lambda X: X.blkval.cpu().numpy()
    L = dev(S.project(g["S"]))
    chordal.cholesky(L)
    assert rel(S.dense(host(L), symmetric=False), g["chol"]) < TOL
    Y = L.copy()
    chordal.projected_inverse(Y)
    assert rel(S.dense(host(Y)), g["projinv"]) < TOL
    Lc = Y.copy()
    chordal.completion(Lc)
    assert rel(S.dense(host(Lc), symmetric=False), g["chol"]) < 1e-9
    U = dev(S.project(g["U"]))
    chordal.hessian(L, Y, U, adj=None, inv=False)
    assert rel(S.dense(host(U)), g["hessU"]) < TOL
    chordal.hessian(L, Y, U, adj=None, inv=True)
    assert rel(S.dense(host(U)), g["U"]) < 1e-9
    cptr, cidx, cval = constraints(symb, g["A"])
    m = len(cptr) - 1
    for tnz in (0.0, 0.9):                                 # Hessian (Gram) path and column-sparse path of the Schur complement
        sys_ = KKTSystem(symb, cptr, cidx, cval, max_rhs=2, tnzcols=tnz)
        solve = sys_.factor(L, Y)
        assert rel(np.tril(sys_.H.cpu().numpy().T), np.linalg.cholesky(g["schur"])) < 1e-9
        bx, by = dev(S.project(g["bx"])), torch.from_numpy(g["by"].copy()).cuda()
        solve(bx, by, g["kk"])
        assert rel(S.dense(host(bx)), g["x"]) < 1e-9 and rel(by.cpu().numpy(), g["y"]) < 1e-9

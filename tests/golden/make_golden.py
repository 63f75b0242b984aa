#!/usr/bin/env python3
"""Generates the fixtures tests/golden/*.npz.

What they are: DENSE numpy restatements of the mathematical definitions of the calls on the Newton-KKT path
(reference call sites src/python/solvers.py:479-541, 881-891), evaluated in extended
precision-free plain float64 dense linear algebra on small chordal patterns.  Neither the CPU oracle
(oracle/chordal_oracle.c) nor the HIP library takes part in producing them, so both can be checked against them.

What they are NOT: outputs of the reference.  CHOMPACK / CVXOPT are not installable here, the
reference holds no golden vectors for this path, so parity with the reference itself stays unpinned.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from smcp_amd import problems  # noqa: E402  (pattern generators only: plain numpy)

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (pattern, m)   -- natural order is a perfect elimination order for all of them (cliques children first)
    "band_n30_bw3": (lambda: problems.band_pattern(30, 3), 5),                       # config-1 shape at test size
    "arrow_6x4_5": (lambda: problems.block_arrow_pattern(6, 4, 5), 4),               # config-3 shape at test size
    "nested_small": (lambda: problems.nested_block_arrow_pattern(nsub=2, nmid=3, nleaf_per_mid=2, leaf=(2, 4),
                                                                 mid=(3, 5), top=(4, 6), root=8, seed=1), 6),  # config-5 shape
    "diag_n12": (lambda: problems.band_pattern(12, 0), 3),                           # LP case
}


def mask_of(pat):
    n, cp, ri = pat
    M = np.zeros((n, n), dtype=bool)
    for j in range(n):
        M[ri[cp[j]:cp[j + 1]], j] = True
    M |= M.T
    M[np.diag_indices(n)] = True
    return M


def is_peo_without_fill(M):
    """Symbolic Cholesky in natural order produces no fill <=> natural order is a PEO of a chordal pattern."""
    n = M.shape[0]
    F = np.tril(M).copy()
    for j in range(n):
        rows = np.nonzero(F[j + 1:, j])[0] + j + 1
        for a in rows:
            F[rows[rows >= a], a] = True
    return (np.tril(M) == F).all()


def make_case(name, pat, m, seed):
    rng = np.random.default_rng(seed)
    n = pat[0]
    V = mask_of(pat)
    assert is_peo_without_fill(V), name
    proj = lambda A: np.where(V, A, 0.0)
    # S positive definite with pattern exactly V: L0 L0^T with L0 lower on V (zero fill: V chordal, natural order PEO)
    L0 = np.where(np.tril(V), rng.standard_normal((n, n)) * 0.4, 0.0)
    L0[np.diag_indices(n)] = 1.0 + rng.random(n)
    S = L0 @ L0.T
    assert (np.abs(S[~V]) < 1e-14).all()
    Lc = np.linalg.cholesky(S)                       # cholesky(S)            (A.2)
    Si = np.linalg.inv(S)
    Y = proj(Si)                                     # projected_inverse      (A.3): P_V(S^-1)
    # completion (A.4): for X = P_V(S^-1) the maximum-determinant completion is S^-1, so the factor returned is Lc
    hess = lambda U: proj(Si @ U @ Si)               # hessian(L, Y, U, adj=None): P_V(S^-1 U S^-1)   (A.5)
    sym = lambda A: proj(A + A.T)
    U1 = sym(rng.standard_normal((n, n)))
    A = np.stack([sym(rng.standard_normal((n, n)) * (rng.random((n, n)) < 0.35)) for _ in range(m)])
    for i in range(m):
        A[i][np.diag_indices(n)] += 0.1 * (i + 1)
    # Schur complement of kkt_chol (solvers.py:479-497): H_ij = <A_i, hessian(A_j)> = tr(A_i S^-1 A_j S^-1)
    H = np.array([[np.sum(A[i] * hess(A[j])) for j in range(m)] for i in range(m)])
    # solve_ (solvers.py:506-541):  [-kk Hess^-1  A^adj ; A  0] [x; y] = [bx; by],  x in S_V
    bx = sym(rng.standard_normal((n, n)))
    by = rng.standard_normal(m)
    kk = 0.37
    Amap = lambda X: np.array([np.sum(A[i] * X) for i in range(m)])
    Aadj = lambda y: np.tensordot(y, A, axes=1)
    y = np.linalg.solve(H, kk * by + Amap(hess(bx)))
    x = hess(Aadj(y) - bx) / kk
    # independent verification of the defining equations: Hess^-1 by solving the |V|-dimensional linear system
    idx = np.argwhere(np.tril(V))
    nv = len(idx)

    def vec(M):
        return np.array([M[i, j] for i, j in idx])

    def unvec(v):
        M = np.zeros((n, n))
        for (i, j), t in zip(idx, v):
            M[i, j] = t
            M[j, i] = t
        return M

    Hm = np.stack([vec(hess(unvec(e))) for e in np.eye(nv)], axis=1)
    hinv_x = unvec(np.linalg.solve(Hm, vec(x)))
    r1 = -kk * hinv_x + Aadj(y) - bx
    r2 = Amap(x) - by
    assert np.abs(proj(r1)).max() < 1e-8 * (1 + np.abs(bx).max()) and np.abs(r2).max() < 1e-8 * (1 + np.abs(by).max())
    n_, cp, ri = pat
    np.savez_compressed(os.path.join(HERE, name + ".npz"), n=n_, colptr=cp, rowind=ri, S=S, chol=Lc, projinv=Y,
                        U=U1, hessU=hess(U1), A=A, schur=H, bx=bx, by=by, kk=kk, x=x, y=y, hinv_x=hinv_x)
    print("%-14s n=%3d |V|=%4d m=%d  cond(S)=%.1e cond(H)=%.1e" % (name, n, nv, m, np.linalg.cond(S), np.linalg.cond(H)))


if __name__ == "__main__":
    for s, (name, (mk, m)) in enumerate(sorted(CASES.items())):
        make_case(name, mk(), m, 100 + s)

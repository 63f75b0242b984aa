"""Pins the CPU oracle (oracle/chordal_oracle.c) with exact dense identities.

No golden vectors exist for this path (CHOMPACK is absent from the reference tree and the
reference's one test asserts nothing), so the oracle is checked against dense numpy linear
algebra on small chordal patterns; tolerance 1e-10 relative (fp64, well-conditioned inputs).
The oracle is run on BOTH the product's symbolic arrays and the pure-Python reference symbolic.
"""
import numpy as np
import pytest

from oracle import oracle as orc
from oracle.symbolic_ref import symbolic_ref
from tests.helpers import PATTERNS, edges_of, make, proj, random_spd_on_V

TOL = 1e-10
NAMES = sorted(PATTERNS)


@pytest.fixture(autouse=True, params=["loops", "blas"])
def _dense_backend(request):
    """Every identity runs twice: on the plain loops (the parity checker) and with the per-clique dense operations
    routed through the host BLAS / LAPACK (the cpu_baseline path of bench.py), here from dimension 2 on so that the
    small test patterns exercise every BLAS call."""
    if request.param == "blas":
        if orc.use_blas(True, min_dim=2) is None:
            pytest.skip("no host BLAS available through scipy")
    yield
    orc.use_blas(False)


def rel(a, b):
    return np.linalg.norm(a - b) / max(1.0, np.linalg.norm(b))


def syms(name):
    pat, symb, S = make(name)
    yield S
    yield orc.Sym(symbolic_ref(pat[0], edges_of(pat)))


@pytest.mark.parametrize("name", NAMES)
def test_cholesky_llt(name):
    for S in syms(name):
        A, L = random_spd_on_V(S, 1)
        x = S.project(A)
        orc.cholesky(S, x)
        Ld = S.dense(x, symmetric=False)
        assert rel(Ld @ Ld.T, A) < TOL                 # L L^T = X exactly, also OFF V (zero fill)
        assert rel(Ld, L) < 1e-8
        assert abs(orc.logdiagsum(S, x) - np.log(np.diag(L)).sum()) < 1e-9
        orc.llt(S, x)
        assert rel(S.dense(x), A) < TOL


@pytest.mark.parametrize("name", NAMES)
def test_cholesky_rejects_indefinite(name):
    for S in syms(name):
        A, _ = random_spd_on_V(S, 2)
        A[S.n // 2, S.n // 2] = -1.0
        with pytest.raises(ArithmeticError):
            orc.cholesky(S, S.project(A))


@pytest.mark.parametrize("name", NAMES)
def test_projected_inverse_and_completion(name):
    for S in syms(name):
        A, _ = random_spd_on_V(S, 3)
        x = S.project(A)
        orc.cholesky(S, x)
        Lfac = x.copy()
        orc.projected_inverse(S, x)
        Y = proj(S, np.linalg.inv(A))
        assert rel(S.dense(x), Y) < TOL                # Y = P_V(S^-1)
        orc.completion(S, x)                           # round trip: returns the factor of S
        assert rel(S.dense(x, False), S.dense(Lfac, False)) < 1e-8
        Ld = S.dense(x, False)
        assert rel(proj(S, np.linalg.inv(Ld @ Ld.T)), Y) < TOL   # P_V((L L^T)^-1) = X


@pytest.mark.parametrize("name", NAMES)
def test_completion_rejects_noncompletable(name):
    for S in syms(name):
        A, _ = random_spd_on_V(S, 4)
        X = proj(S, np.linalg.inv(A))
        X[0, 0] = -abs(X[0, 0])
        with pytest.raises(ArithmeticError):
            orc.completion(S, S.project(X))


@pytest.mark.parametrize("name", NAMES)
def test_hessian_identities(name):
    for S in syms(name):
        rng = np.random.default_rng(5)
        A, _ = random_spd_on_V(S, 5)
        Ai = np.linalg.inv(A)
        L = S.project(A)
        orc.cholesky(S, L)
        Y = L.copy()
        orc.projected_inverse(S, Y)
        U = rng.standard_normal((S.n, S.n))
        U = proj(S, U + U.T)
        V = rng.standard_normal((S.n, S.n))
        V = proj(S, V + V.T)
        u0, v0 = S.project(U), S.project(V)
        # full Hessian: P_V(S^-1 U S^-1)
        u = u0.copy()
        orc.hessian(S, L, Y, u, adj=None, inv=False)
        HU = proj(S, Ai @ U @ Ai)
        assert rel(S.dense(u), HU) < TOL
        # inverse undoes it
        orc.hessian(S, L, Y, u, adj=None, inv=True)
        assert rel(u, u0) < 1e-9
        # factors: H = G^adj o G ; <GU,GU> = tr(S^-1 U S^-1 U) ; <GU,V> = <U,G^adj V>
        g = u0.copy()
        orc.hessian(S, L, Y, g, adj=False, inv=False)
        assert abs(orc.dot(S, g, g) - np.trace(Ai @ U @ Ai @ U)) < 1e-9 * max(1, abs(np.trace(Ai @ U @ Ai @ U)))
        ga = v0.copy()
        orc.hessian(S, L, Y, ga, adj=True, inv=False)
        assert abs(orc.dot(S, g, v0) - orc.dot(S, u0, ga)) < 1e-9 * max(1, abs(orc.dot(S, g, v0)))
        gg = g.copy()
        orc.hessian(S, L, Y, gg, adj=True, inv=False)
        assert rel(S.dense(gg), HU) < TOL
        # factor inverses
        gi = g.copy()
        orc.hessian(S, L, Y, gi, adj=False, inv=True)
        assert rel(gi, u0) < 1e-9
        gai = ga.copy()
        orc.hessian(S, L, Y, gai, adj=True, inv=True)
        assert rel(gai, v0) < 1e-9


@pytest.mark.parametrize("name", NAMES)
def test_trsm_dot(name):
    for S in syms(name):
        rng = np.random.default_rng(6)
        A, Ld = random_spd_on_V(S, 6)
        L = S.project(A)
        orc.cholesky(S, L)
        Ld = S.dense(L, False)
        B = rng.standard_normal((3, S.n))
        b = B.copy()
        orc.trsm(S, L, b, "N")
        assert rel(b.T, np.linalg.solve(Ld, B.T)) < TOL
        b = B.copy()
        orc.trsm(S, L, b, "T")
        assert rel(b.T, np.linalg.solve(Ld.T, B.T)) < TOL
        X = proj(S, rng.standard_normal((S.n, S.n)))
        X = proj(S, X + X.T)
        assert abs(orc.dot(S, S.project(X), S.project(A)) - np.trace(X @ A)) < 1e-10 * max(1, abs(np.trace(X @ A)))


@pytest.mark.parametrize("name", ["arrow", "rand2"])
def test_kkt_solve_residuals(name):
    """kkt_chol + solve_ restated around the oracle: residuals of solvers.py:401-411 below 1e-10
    (the reference's own DEBUG check, solvers.py:534-538)."""
    from smcp_amd import problems
    pat, symb, S = make(name)
    rng = np.random.default_rng(7)
    A, _ = random_spd_on_V(S, 7)
    L = S.project(A)
    orc.cholesky(S, L)
    Y = L.copy()
    orc.projected_inverse(S, Y)
    m = 5
    cptr, cidx, cval = problems.random_constraints(symb, m, density=0.2, seed=3)
    K = orc.KKT(S, cptr, cidx, cval)
    H = K.schur_factor(L, Y)
    # H_ij = tr(A_i S^-1 A_j S^-1) before factoring
    Ai = np.linalg.inv(A)
    Ad = [S.dense(K.constraint(j)) for j in range(m)]
    Href = np.array([[np.trace(Ad[i] @ Ai @ Ad[j] @ Ai) for j in range(m)] for i in range(m)])
    Hl = np.tril(H)
    assert rel(Hl @ Hl.T, Href) < 1e-9
    bxd = rng.standard_normal((S.n, S.n))
    bx = S.project(proj(S, bxd + bxd.T))
    by = rng.standard_normal(m)
    for kk in (1.0, 0.37):
        x, y = K.solve(L, Y, H, bx, by, kk)
        r, rr = K.residual(L, Y, x, y, bx, by, kk)
        assert np.sqrt(orc.dot(S, r, r)) / max(1, np.sqrt(orc.dot(S, bx, bx))) < 1e-10
        assert np.linalg.norm(rr) / max(1, np.linalg.norm(by)) < 1e-10


@pytest.mark.parametrize("name", ["arrow", "rand2"])
def test_kkt_qr_restatement(name):
    """kkt_qr (solvers.py:413-475) restated around the oracle: R^T R = H / 2 (the svec scaling of solvers.py:420),
    the solution agrees with kkt_chol's and passes the reference's DEBUG residual check (solvers.py:465-469)."""
    from smcp_amd import problems
    pat, symb, S = make(name)
    rng = np.random.default_rng(17)
    A, _ = random_spd_on_V(S, 17)
    L = S.project(A)
    orc.cholesky(S, L)
    Y = L.copy()
    orc.projected_inverse(S, Y)
    m = 5
    cptr, cidx, cval = problems.random_constraints(symb, m, density=0.2, seed=13)
    K = orc.KKT(S, cptr, cidx, cval)
    H = K.schur_factor(L, Y)
    Hl = np.tril(H)
    F = K.qr_factor(L, Y)
    assert rel(F["R"].T @ F["R"], 0.5 * (Hl @ Hl.T)) < 1e-10
    bxd = rng.standard_normal((S.n, S.n))
    bx = S.project(proj(S, bxd + bxd.T))
    by = rng.standard_normal(m)
    for kk in (1.0, 0.37):
        xc, yc = K.solve(L, Y, H, bx, by, kk)
        x, y = K.qr_solve(L, Y, F, bx, by, kk)
        assert rel(y, yc) < 1e-10
        assert rel(x, xc) < 1e-10
        r, rr = K.residual(L, Y, x, y, bx, by, kk)
        assert np.sqrt(orc.dot(S, r, r)) / max(1, np.sqrt(orc.dot(S, bx, bx))) < 1e-10
        assert np.linalg.norm(rr) / max(1, np.linalg.norm(by)) < 1e-10

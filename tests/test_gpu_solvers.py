"""End-to-end interior-point runs on the GPU path (config 1 plumbing + the reference's own test problem).

The reference's test (tests/test_basic.py:6-22) asserts nothing, so optimality is certified here
independently: primal/dual feasibility and duality gap recomputed in dense numpy from the returned
solution (DIMACS-style errors, doc benchmarks index.rst:113-122)."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def _certify(P, sol, tol=1e-6):
    n, m = P.n, P.m
    X = np.asarray(sol["x"].todense())
    S = np.asarray(sol["s"].todense())
    y = sol["y"]
    C = np.asarray(P.get_A(0).todense())
    Ai = [np.asarray(P.get_A(i + 1).todense()) for i in range(m)]
    pres = np.linalg.norm(np.array([np.sum(a * X) for a in Ai]) - P.b) / (1 + np.abs(P.b).max())
    dres = np.linalg.norm(sum(yi * a for yi, a in zip(y, Ai)) + S - C) / (1 + np.abs(C).max())
    assert pres < tol and dres < tol
    assert np.linalg.eigvalsh(S).min() > -1e-8                     # S in the PSD cone
    pc, dc = np.sum(C * X), float(P.b @ y)
    assert abs(pc - dc) / (1 + abs(pc) + abs(dc)) < 1e-5
    assert abs(np.sum(X * S)) / (1 + abs(pc)) < 1e-5                # complementarity
    return pc, dc


@pytest.fixture(params=["mcs", "auto"])
def peo(request, _reference_elimination_order):
    """the suite pins options['peo'] = 'mcs' (the stored histories were generated on the reference's ordering
    sequence); the tests that compare optimum, status and certificates rather than 1e-9 histories also run on the order users
    get by default ('auto': the given order when it has zero fill -- another clique tree, other level counts and kernel shape
    classes)."""
    from smcp_amd import solvers
    solvers.options["peo"] = request.param
    return request.param


def _starts(P):
    return ({"x": sp.csc_matrix(np.tril(P._X0))}, {"y": P._y0, "s": sp.csc_matrix(np.tril(P._S0))})


def test_band_sdp_config1_feas_primal_and_dual_scaling(peo):
    """BASELINE config 1: band SDP n=200, half-bandwidth 3, m=100 through the feasible-start
    driver (the reference's benchmark method M1: 36-38 iterations, DIMACS feasibility errors
    ~1e-16, doc band_ex1_*.html)."""
    from smcp_amd import base, solvers
    solvers.options.update(show_progress=False, maxiters=100)
    P = base.band_SDP(200, 100, 3, seed=0)
    assert P.nnz == 794
    ps, ds = _starts(P)
    sol_p = P.solve_feas(scaling="primal", primalstart=ps, dualstart=ds)
    assert sol_p["status"] == "optimal" and sol_p["iterations"] <= 60
    pc, dc = _certify(P, sol_p)
    assert max(abs(sol_p["dimacs"][0]), abs(sol_p["dimacs"][2])) < 1e-12     # feasible-start stays feasible
    # X is PSD-completable: its maximal cliques are PSD
    X = np.asarray(sol_p["x"].todense())
    for j in range(0, 196, 17):
        assert np.linalg.eigvalsh(X[j:j + 4, j:j + 4]).min() > -1e-8
    sol_d = P.solve_feas(scaling="dual", primalstart=ps, dualstart=ds)
    assert sol_d["status"] == "optimal"
    pc2, dc2 = _certify(P, sol_d)
    assert abs(pc - pc2) < 1e-5 * (1 + abs(pc))
    assert pc <= np.sum(np.asarray(P.get_A(0).todense()) * P._X0) + 1e-6    # the start bounds the optimum


def test_band_sdp_config1_via_conelp(peo):
    """Config 1 "via smcp.solvers.conelp": the same band SDP in CVXOPT cone-LP form (dims s=[n]), which runs
    the self-dual-embedding driver, at the DEFAULT tolerances (feastol 1e-8, abstol/reltol 1e-6), both scalings
    through the embedding driver; the optimum must agree with the feasible-start solver's."""
    from smcp_amd import base, solvers
    solvers.options.update(show_progress=False, maxiters=100, feastol=1e-8, abstol=1e-6, reltol=1e-6)
    P = base.band_SDP(200, 100, 3, seed=0)
    n, m = P.n, P.m
    G = sp.hstack([sp.csc_matrix(P.get_A(i + 1).reshape((n * n, 1), order="F")) for i in range(m)]).tocsc()
    h = np.asarray(P.get_A(0).todense()).reshape(-1, order="F")
    sol = solvers.conelp(-P.b, G, h, {"l": 0, "q": [], "s": [n]})
    assert sol["status"] == "optimal" and sol["iterations"] <= 40
    assert sol["primal infeasibility"] <= 1e-8 and sol["dual infeasibility"] <= 1e-8
    ref = P.solve_feas(scaling="dual", primalstart=_starts(P)[0], dualstart=_starts(P)[1])
    assert abs(sol["dual objective"] - ref["dual objective"]) < 1e-5 * (1 + abs(ref["dual objective"]))
    assert abs(sol["primal objective"] - ref["dual objective"]) < 1e-5 * (1 + abs(ref["dual objective"]))
    for sc in ("primal", "dual"):
        se = P.solve_esd(scaling=sc)
        assert se["status"] == "optimal" and se["iterations"] <= 40
        _certify(P, se)


def test_reference_conelp_example(peo):
    """The problem of the reference's tests/test_basic.py (CVXOPT manual example, l=2, q=[4,4], s=[3])."""
    from smcp_amd import solvers
    solvers.options["show_progress"] = False
    solvers.options["maxiters"] = 100
    c = np.array([-6., -4., -5.])
    G = np.array([[16., 7., 24., -8., 8., -1., 0., -1., 0., 0., 7., -5., 1., -5., 1., -7., 1., -7., -4.],
                  [-14., 2., 7., -13., -18., 3., 0., 0., -1., 0., 3., 13., -6., 13., 12., -10., -6., -10., -28.],
                  [5., 0., -15., 12., -6., 17., 0., 0., 0., -1., 9., 6., -6., 6., -7., -7., -6., -7., -11.]]).T
    h = np.array([-3., 5., 12., -2., -14., -13., 10., 0., 0., 0., 68., -30., -19., -30., 99., 23., -19., 23., 10.])
    dims = {"l": 2, "q": [4, 4], "s": [3]}
    solvers.options.update(feastol=1e-8, abstol=1e-6, reltol=1e-6)      # the defaults
    sol = solvers.conelp(c, G, h, dims)
    assert sol["status"] == "optimal"
    x, s, z = sol["x"], sol["s"], sol["z"]
    assert np.linalg.norm(G @ x + s - h) < 1e-6 * (1 + np.linalg.norm(h))      # primal feasibility
    assert np.linalg.norm(G.T @ z + c) < 1e-6 * (1 + np.linalg.norm(c))        # dual feasibility
    assert abs(c @ x + h @ z) < 1e-5 * (1 + abs(c @ x))                        # zero duality gap
    # cone membership of the slack and of the multiplier
    for v in (s, z):
        assert (v[:2] > -1e-8).all()
        assert v[2] >= np.linalg.norm(v[3:6]) - 1e-7 and v[6] >= np.linalg.norm(v[7:10]) - 1e-7
        M = v[10:].reshape(3, 3)
        assert np.linalg.eigvalsh((M + M.T) / 2).min() > -1e-7
    # CVXOPT manual's optimum for this example
    assert np.allclose(x, [-1.22, 0.0966, 3.58], atol=5e-3)


def test_lp_against_scipy(peo):
    """LP through the LP -> diagonal-SDP mapping (solvers.py:2505-2509) against scipy.optimize.linprog."""
    from scipy.optimize import linprog
    from smcp_amd import solvers
    solvers.options["show_progress"] = False
    rng = np.random.default_rng(0)
    nvar, ncon = 6, 15
    G = rng.standard_normal((ncon, nvar))
    x0 = rng.standard_normal(nvar)
    h = G @ x0 + rng.random(ncon) + 0.1
    z0 = rng.random(ncon) + 0.1
    c = -G.T @ z0
    # the embedding's endgame hovers around 1e-8 (identical iterates with the CPU oracle backend up to
    # that point), so the feasibility tolerance is stated one decade above the default
    solvers.options["feastol"] = 1e-7
    try:
        sol = solvers.lp(c, G, h)
    finally:
        solvers.options["feastol"] = 1e-8
    ref = linprog(c, A_ub=G, b_ub=h, bounds=[(None, None)] * nvar, method="highs")
    assert sol["status"] == "optimal" and ref.status == 0
    assert abs(c @ sol["x"] - ref.fun) < 1e-5 * (1 + abs(ref.fun))


def test_maxcut_config4_shape(peo):
    """BASELINE config 4 at test size: max-cut SDP on a random non-chordal graph (n = 300, 900 edges; the
    full G51-sized instance n = 1000 / 5909 edges runs in scratch/maxcut.py: optimal in 31 iterations).
    The symbolic layer embeds the pattern with its own minimum-degree ordering; optimality is certified in
    dense numpy (diag(X) = 1, Diag(y) + S = C, S >= 0, zero gap)."""
    from smcp_amd import base, solvers
    solvers.options.update(show_progress=False, maxiters=60)
    P = base.maxcut_SDP(300, 900, seed=0)
    assert not P.ischordal
    C = P.get_A(0)
    y0 = -np.ones(300) * (abs(C).sum(axis=1).max() + 1.0)
    sol = P.solve_feas(scaling="dual", dualstart={"y": y0})
    assert sol["status"] == "optimal"
    X = np.asarray(sol["x"].todense())
    S = np.asarray(sol["s"].todense())
    Cd = np.asarray(C.todense())
    assert np.allclose(np.diag(X), 1.0, atol=1e-7)
    assert np.linalg.norm(np.diag(sol["y"]) + S - Cd) < 1e-7 * (1 + np.abs(Cd).max())
    assert np.linalg.eigvalsh(S).min() > -1e-7
    assert abs(np.sum(Cd * X) - sol["y"].sum()) < 1e-5 * (1 + abs(sol["y"].sum()))


def test_maxcut_config4_full_size_through_sdpa_file(tmp_path):
    """BASELINE config 4 at full size: max-cut relaxation on a random 1000-node / 5909-edge graph (the size of SDPLIB's
    maxG51; the file itself cannot be obtained offline), m = 1000 column-sparse constraints, non-chordal pattern embedded
    by the library's own minimum-degree ordering.  The problem takes the route a real SDPLIB file would: written in SDPA
    sparse format, read back through SDP(filename) (N1), solved by the feasible-start driver with dual scaling.  Dense
    dual certificate: S = C - Diag(y) is positive semidefinite and b'y equals the primal objective; diag(X) = 1."""
    from smcp_amd import base, solvers
    solvers.options.update(show_progress=False, maxiters=80)
    n = 1000
    P0 = base.maxcut_SDP(n, 5909, seed=0)
    P0.write_sdpa(str(tmp_path / "maxcut_g51like"))
    P = base.SDP(str(tmp_path / "maxcut_g51like.dat-s"))
    assert (P.n, P.m) == (n, n) and not P.ischordal
    assert abs(sp.csc_matrix(P.A) - sp.csc_matrix(P0.A)).max() == 0 and np.array_equal(P.b, P0.b)
    C = P.get_A(0)
    y0 = -np.ones(n) * (abs(C).sum(axis=1).max() + 1.0)
    sol = P.solve_feas(scaling="dual", dualstart={"y": y0})
    assert sol["status"] == "optimal"
    Cd = np.asarray(C.todense())
    y = np.asarray(sol["y"]).reshape(-1)
    Sd = Cd - np.diag(y)
    assert np.linalg.eigvalsh(Sd).min() > -1e-6 * (1 + np.abs(Cd).max())          # dual feasible
    X = sol["x"]
    assert np.allclose(X.diagonal(), 1.0, atol=1e-6)                              # primal feasible on the constraints
    pobj = float(C.multiply(X).sum())
    assert abs(pobj - y.sum()) < 1e-5 * (1 + abs(pobj))                            # zero duality gap
    assert max(abs(sol["dimacs"][0]), abs(sol["dimacs"][2])) < 1e-5


def test_phase1_finds_strictly_feasible_point():
    """Row N3: SDP.solve_phase1 (base.py:370-470, misc.phase1_sdp): when the least-norm solution of the
    equality constraints is not positive definite, the Phase-I SDP solved by the feasible-start driver must
    return a strictly feasible primal point, which then starts the main problem."""
    from smcp_amd import base, chordal, solvers
    solvers.options.update(show_progress=False, maxiters=100)
    P = base.band_SDP(60, 20, 2, seed=4)
    X0, sol1 = P.solve_phase1()
    assert X0 is not None
    assert sol1 is not None and sol1["status"] == "optimal"     # the least-norm point was infeasible: Phase-I SDP solved
    n = P.n
    Xd = np.asarray(X0.todense())
    for i in range(P.m):                                    # <A_i, X0> = b_i
        Ai = np.asarray(P.get_A(i + 1).todense())
        assert abs(np.sum(Ai * Xd) - P.b[i]) < 1e-6 * (1 + abs(P.b[i]))
    Pr = solvers._Problem(P.A, P.b)
    Xc = Pr.from_sym(X0)
    chordal.completion(Xc)                                  # strictly inside the cone of PSD-completable matrices
    sol = P.solve_feas(primalstart={"x": X0}, scaling="primal")
    assert sol["status"] == "optimal"
    ref = P.solve_feas(scaling="dual", primalstart=_starts(P)[0], dualstart=_starts(P)[1])
    assert abs(sol["primal objective"] - ref["primal objective"]) < 1e-4 * (1 + abs(ref["primal objective"]))


def test_dense_completion_and_mtxnorm_on_device():
    """smcp.completion (dense maximum-determinant completion, base.py:952-973) and a matrix-norm SDP
    (base.py:639-773) through the HIP path."""
    import smcp_amd
    from smcp_amd import base, solvers
    rng = np.random.default_rng(0)
    n = 40
    mask = np.abs(np.subtract.outer(np.arange(n), np.arange(n))) <= 3
    Lb = np.where(np.tril(mask), rng.standard_normal((n, n)) * 0.4, 0.0)
    Lb[np.diag_indices(n)] = 1.0 + rng.random(n)
    Si = np.linalg.inv(Lb @ Lb.T)
    Z = smcp_amd.completion(sp.csc_matrix(np.where(np.tril(mask), Si, 0.0)))
    assert np.linalg.norm(Z - Si) < 1e-9 * np.linalg.norm(Si)
    solvers.options.update(show_progress=False, maxiters=100, feastol=1e-8, abstol=1e-6, reltol=1e-6)
    p, q, r = 12, 4, 6
    P = base.mtxnorm_SDP(p, q, r, seed=1)
    sol = P.solve_esd()
    assert sol["status"] == "optimal"
    y = sol["y"]
    nn = p + q
    blk = lambda col: np.asarray(P.A[:, col].todense()).reshape((nn, nn), order="F")[q:, :q]
    M = blk(0) - sum(y[i] * blk(i + 1) for i in range(r))
    assert abs(np.linalg.norm(M, 2) - y[r]) < 1e-5 * (1 + y[r])


@pytest.mark.parametrize("n,m,bw", [(10, 1, 2), (3, 1, 1), (2, 1, 0), (6, 3, 5), (40, 5, 0)])
def test_edge_sizes_through_all_drivers(n, m, bw):
    """Smallest and degenerate shapes: one constraint, diagonal (LP) patterns, a single dense clique; every driver
    reaches the default tolerances and primal and dual objectives agree."""
    from smcp_amd import base, solvers
    solvers.options.update(show_progress=False, maxiters=100, feastol=1e-8, abstol=1e-6, reltol=1e-6)
    P = base.band_SDP(n, m, bw, seed=1)
    st = dict(primalstart={"x": P._X0}, dualstart={"y": P._y0, "s": P._S0})
    sols = [P.solve_feas(scaling="primal", **st), P.solve_feas(scaling="dual", **st), P.solve_esd()]
    for sol in sols:
        assert sol["status"] == "optimal"
        assert abs(sol["primal objective"] - sol["dual objective"]) < 1e-4 * (1 + abs(sol["dual objective"]))
    assert abs(sols[0]["primal objective"] - sols[2]["primal objective"]) < 1e-4 * (1 + abs(sols[2]["primal objective"]))


def test_one_by_one_sdp():
    """n = 1: minimize 2 x subject to x = 3, x >= 0."""
    from smcp_amd import base, solvers
    solvers.options.update(show_progress=False, maxiters=100)

    class One(base.SDP):
        def __init__(self):
            super().__init__()
            self._A = sp.csc_matrix(np.array([[2.0, 1.0]]))
            self._b = np.array([3.0])
            self._blockstruct = [1]

    sol = One().solve_esd()
    assert sol["status"] == "optimal" and abs(sol["primal objective"] - 6.0) < 1e-5 and abs(sol["dual objective"] - 6.0) < 1e-5


@pytest.mark.gpu
def test_kktsolver_qr_on_device_matches_chol():
    from smcp_amd import base
    """kktsolver='qr' end to end on the device (feasible-start and embedding drivers, a problem with column-sparse
    constraints so that the constraint classification has to be redone): same optimum and iteration count as
    'chol', primal feasibility at least as good."""
    P = base.band_SDP(60, 20, 3, seed=5)
    fc = P.solve_feas(kktsolver="chol")
    fq = P.solve_feas(kktsolver="qr")
    eq = P.solve_esd(kktsolver="qr")
    for sol in (fc, fq, eq):
        assert sol["status"] == "optimal"
    assert abs(fq["primal objective"] - fc["primal objective"]) < 1e-6 * (1 + abs(fc["primal objective"]))
    assert abs(eq["primal objective"] - fc["primal objective"]) < 1e-5 * (1 + abs(fc["primal objective"]))
    assert abs(fq["iterations"] - fc["iterations"]) <= 1
    assert fq["primal infeasibility"] < 1e-8


def test_omega_neighbourhood_linesearch_on_device():
    """options['eta']: bisection on Omega(X, S) for the tangent step (solvers.py:662-689), on the device."""
    from smcp_amd import base, solvers
    P = base.band_SDP(60, 20, 3, seed=6)
    saved = dict(solvers.options)
    try:
        solvers.options.update(show_progress=False)
        ref = P.solve_feas()
        solvers.options["eta"] = 5.0
        sol = P.solve_feas()
    finally:
        solvers.options.clear()
        solvers.options.update(saved)
    assert sol["status"] == "optimal" and ref["status"] == "optimal"
    assert abs(sol["primal objective"] - ref["primal objective"]) < 1e-5 * (1 + abs(ref["primal objective"]))
    _certify(P, sol)


def test_ipm_golden_cases_on_device():
    """The same ten runs on the device: same optimum (the iterates differ in the last bits, so the step counts may
    differ by one and the objectives by the stopping tolerance).  This test found kkt_qr_solve using a stale
    chol(Y_AA) cache after a line-search completion (36 iterations against 34 before the fix)."""
    from smcp_amd import solvers
    import ipm_golden
    saved = dict(solvers.options)
    try:
        solvers.options.update(show_progress=False, maxiters=100)
        ipm_golden.check_all(iter_slack=1, obj_tol=2e-6, y_tol=2e-4)
    finally:
        solvers.options.clear()
        solvers.options.update(saved)


def test_socp_and_sdp_front_ends_on_device():
    """solvers.socp / solvers.sdp (solvers.py:2608-2699) return the reference's split keys; same cases and
    independent checks as the CPU suite (tests/test_host_solvers.py), here on the HIP path."""
    from smcp_amd import solvers
    from tests import test_host_solvers as th
    solvers.options.update(show_progress=False, maxiters=100)
    case = th._socp_case()
    th.check_socp_solution(solvers.socp(*case), *case)
    case = th._sdp_case()
    th.check_sdp_solution(solvers.sdp(*case), *case)


def test_phase1_golden_cases_on_device():
    """Row N3: the Phase-I fixtures generated over the CPU oracle (tests/golden/phase1_cases.json: least-norm branch
    and augmented-SDP branch) are reproduced on the HIP path: same branch, Phase-I iteration count +- 1, X0 feasible
    to 1e-9 and strictly inside the cone, the main problem started from X0 reaches the stored optimum."""
    import ipm_golden
    from smcp_amd import solvers
    solvers.options.update(show_progress=False, maxiters=100)
    ipm_golden.check_phase1(iter_slack=1, obj_tol=2e-6)


def test_random_interior_point_runs_agree_between_kkt_solvers():
    """tests/fuzz_ipm.py on eight random problems (seeds 99008-99015; 99010 is one of the three runs that ended `unknown`
    while fronts without separator took both Hessian sweeps as Y_NN F Y_NN -- DESIGN section 3 "Round 3"): six whole
    interior-point runs per problem, all optimal, chol and qr in step."""
    import fuzz_ipm
    bad = fuzz_ipm.run(16, seed0=99000, first=8)
    assert not bad, bad

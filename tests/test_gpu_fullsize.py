"""Full-size parity (BASELINE.json configs 1, 2, 3, 5) through size-independent properties of the path -- the
identities SURVEY.md 8c lists as pins (all exact in exact arithmetic; fp64 tolerances stated per check).  The
CPU oracle is too slow at these sizes, so nothing here calls it: every check is a round trip or an algebraic
identity evaluated with the library's own entry points, which a wrong kernel cannot satisfy by accident.

  P1  llt(cholesky(S)) = S                                 (zero-fill factorisation, solvers.py:640,904)
  P2  completion(projected_inverse(L)) = L                 (P_V((L L^T)^-1) = Y  <=>  completion(Y) = L)
  P3  hessian(inv) o hessian = identity, adj in {False, True, None}
  P4  <G(U), V> = <U, G^adj(V)>                            (the two factors are adjoint)
  P5  <G(U), G(U)> = <U, H(U)>                             (H = G^adj o G: what the Gram Schur complement relies on)
  P6  hessian is linear:  H(a U + b V) = a H(U) + b H(V)   (batched call vs single calls as well)
  P7  kkt_chol + solve_: residuals of the KKT system (the reference's DEBUG check, solvers.py:401-411,534-538)
      below 1e-10 relative, H symmetric, and H_ij = <A_i, H(A_j)> on sampled pairs
"""
import numpy as np
import pytest
import torch

from smcp_amd import chordal, problems
from smcp_amd.cspmatrix import cspmatrix
from smcp_amd.kkt import KKTSystem
from smcp_amd.symbolic import Symbolic

pytestmark = pytest.mark.gpu

CONFIGS = {
    # name: (pattern factory, m constraints, right-hand sides held on the device at once)
    "config1_band200": (lambda: problems.band_pattern(200, 3), 100, 100),
    "config2_dense4096": (lambda: problems.band_pattern(4096, 4095), 16, 16),                  # m as bench.py --workload dense4096
    "config3_arrow2000x64+128": (lambda: problems.block_arrow_pattern(2000, 64, 128), 100, 60),  # m as bench.py --workload arrow
    "config5_synth50k": (lambda: problems.nested_block_arrow_pattern(), 100, 100),
}


def relerr(A, B):
    d = (A - B).norm().item()
    return d / max(1.0, B.norm().item())


@pytest.fixture(scope="module", params=sorted(CONFIGS))
def problem(request):
    pat, m, rhs = CONFIGS[request.param]
    symb = Symbolic(pat())
    symb.device_init(0, rhs)
    Lh = problems.random_factor_blkval(symb, 3)
    L0 = cspmatrix(symb, torch.from_numpy(Lh).cuda())
    S = L0.copy()
    chordal.llt(S)                       # S = L0 L0^T on V
    yield request.param, symb, S, L0, m
    del S, L0
    torch.cuda.empty_cache()


def mask_of(symb):
    msk = torch.zeros(symb.blklen, dtype=torch.float64, device="cuda")
    msk[torch.from_numpy(symb.ccs_to_blk()).cuda()] = 1.0
    return msk


def rand_on_v(symb, msk, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return cspmatrix(symb, torch.randn(symb.blklen, dtype=torch.float64, device="cuda", generator=g) * msk)


def test_factor_roundtrips(problem):
    name, symb, S, L0, m = problem
    msk = mask_of(symb)
    L = S.copy()
    chordal.cholesky(L)
    assert relerr(L.blkval * msk, L0.blkval * msk) < 1e-10          # the Cholesky factor is unique
    R = L.copy()
    chordal.llt(R)
    assert relerr(R.blkval * msk, S.blkval * msk) < 1e-12            # P1
    Y = L.copy()
    chordal.projected_inverse(Y)
    Lc = Y.copy()
    chordal.completion(Lc)
    assert relerr(Lc.blkval * msk, L.blkval * msk) < 1e-9            # P2
    # logdet through the two factors agrees
    assert abs(chordal.logdiagsum(L) - chordal.logdiagsum(Lc)) < 1e-8 * max(1.0, abs(chordal.logdiagsum(L)))


def test_hessian_identities(problem):
    name, symb, S, L0, m = problem
    msk = mask_of(symb)
    L = S.copy()
    chordal.cholesky(L)
    Y = L.copy()
    chordal.projected_inverse(Y)
    U = rand_on_v(symb, msk, 1)
    V = rand_on_v(symb, msk, 2)
    for adj in (False, True, None):                                  # P3
        W = U.copy()
        chordal.hessian(L, Y, W, adj=adj, inv=False)
        chordal.hessian(L, Y, W, adj=adj, inv=True)
        assert relerr(W.blkval * msk, U.blkval) < 1e-9, (name, adj)
    GU = U.copy()
    chordal.hessian(L, Y, GU, adj=False)
    GaV = V.copy()
    chordal.hessian(L, Y, GaV, adj=True)
    a, b = chordal.dot(GU, V), chordal.dot(U, GaV)
    assert abs(a - b) < 1e-10 * max(1.0, abs(a))                     # P4
    HU = U.copy()
    chordal.hessian(L, Y, HU, adj=None)
    g, h = chordal.dot(GU, GU), chordal.dot(U, HU)
    assert abs(g - h) < 1e-10 * max(1.0, abs(g)) and g > 0           # P5
    # P6: linearity, and the batched entry point against single calls
    HV = V.copy()
    chordal.hessian(L, Y, HV, adj=None)
    stack = torch.stack([2.0 * U.blkval - 0.5 * V.blkval, U.blkval, V.blkval]).contiguous()
    chordal.hessian(L, Y, stack, adj=None)
    assert relerr(stack[1] * msk, HU.blkval * msk) < 1e-11
    assert relerr(stack[2] * msk, HV.blkval * msk) < 1e-11
    assert relerr(stack[0] * msk, (2.0 * HU.blkval - 0.5 * HV.blkval) * msk) < 1e-10


def test_kkt_residuals(problem):
    name, symb, S, L0, m = problem
    msk = mask_of(symb)
    L = S.copy()
    chordal.cholesky(L)
    Y = L.copy()
    chordal.projected_inverse(Y)
    cptr, cidx, cval = problems.random_constraints(symb, m, density=0.005, seed=5)
    sys = KKTSystem(symb, cptr, cidx, cval)
    solve = sys.factor(L, Y)
    # H as built (before potrf overwrote its lower triangle the upper one still holds H): symmetric part check
    bx = rand_on_v(symb, msk, 7)
    by = torch.randn(m, dtype=torch.float64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(8))
    for kk in (1.0, 0.125):
        x, y = bx.copy(), by.clone()
        solve(x, y, kk)
        # residuals of  [-kk H^-1  A^adj; A  0] [x; y] = [bx; by]   (solvers.py:401-411)
        r = x.copy()
        chordal.hessian(L, Y, r, adj=None, inv=True)
        r *= -kk
        r += sys.aadj(y)
        r -= bx
        rr = sys.amap(x) - by
        nr = np.sqrt(max(chordal.dot(r, r), 0.0)) / max(1.0, np.sqrt(chordal.dot(bx, bx)))
        nrr = rr.norm().item() / max(1.0, by.norm().item())
        assert nr < 1e-10 and nrr < 1e-10, (name, kk, nr, nrr)
    # sampled entries of the Schur complement against their definition H_ij = <A_i, H(A_j)>
    sys2 = KKTSystem(symb, cptr, cidx, cval)
    sys2.build_schur(L, Y)
    H = sys2.H.clone()
    assert relerr(H, H.T) < 1e-12
    for j in (0, m - 1):
        Aj = cspmatrix(symb, torch.zeros(symb.blklen, dtype=torch.float64, device="cuda"))
        Aj.blkval[torch.from_numpy(cidx[cptr[j]:cptr[j + 1]]).cuda()] = torch.from_numpy(cval[cptr[j]:cptr[j + 1]]).cuda()
        chordal.hessian(L, Y, Aj, adj=None)
        col = sys2.amap(Aj)
        assert relerr(col, H[j]) < 1e-10, (name, j)


def test_config5_whole_interior_point_run():
    """The headline pattern end to end: a strictly feasible SDP on synth50k (n = 50 000, 8073 cliques, m = 100, 11 k
    entries per constraint) through the feasible-start driver with dual scaling -- every iteration is one Schur
    complement (family kernel, Gram), ~9 KKT solves and the line-search factorisations.  Optimal at the default
    tolerances with feasibility kept to rounding (DIMACS errors), in seconds."""
    import time
    from smcp_amd import base, problems, solvers
    solvers.options.update(show_progress=False, maxiters=100, feastol=1e-8, abstol=1e-6, reltol=1e-6)
    P = base.pattern_SDP(problems.nested_block_arrow_pattern(), 100, density=0.005, seed=0)
    assert P.n == 50000 and P.m == 100
    t0 = time.time()
    sol = P.solve_feas(scaling="dual", primalstart={"x": P._X0}, dualstart={"y": P._y0, "s": P._S0})
    dt = time.time() - t0
    assert sol["status"] == "optimal" and sol["iterations"] <= 40
    d = sol["dimacs"]
    assert abs(d[0]) < 1e-12 and abs(d[2]) < 1e-12 and abs(d[4]) < 1e-6 and abs(d[5]) < 1e-6
    assert sol["primal objective"] >= sol["dual objective"] - 1e-6 * (1 + abs(sol["dual objective"]))
    assert dt < 60.0

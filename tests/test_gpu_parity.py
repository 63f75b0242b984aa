"""GPU parity: every C-ABI kernel against the CPU oracle on the same seeded inputs.

Tolerance: fp64, relative 1e-10 on well-conditioned inputs (the north star's "stated fp64
tolerance"); the factorisations differ from the oracle only in summation order.
"""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from smcp_amd import chordal, problems
from smcp_amd.cspmatrix import cspmatrix
from smcp_amd.kkt import KKTSystem
from smcp_amd.symbolic import Symbolic
from tests.helpers import PATTERNS, proj, random_spd_on_V

pytestmark = pytest.mark.gpu
TOL = 1e-10


def rel(a, b):
    """relative error in the 2-norm (absolute only when the expected output is exactly zero)"""
    nb = np.linalg.norm(b)
    return np.linalg.norm(a - b) / nb if nb > 0 else np.linalg.norm(a)


def dev(symb, x):
    return cspmatrix(symb, torch.from_numpy(np.ascontiguousarray(x)).cuda())


def host(X):
    return X.blkval.cpu().numpy()


def lowmask(symb):
    m = np.zeros(symb.blklen, dtype=bool)
    m[symb.ccs_to_blk()] = True
    return m


GPU_PATTERNS = dict(PATTERNS)
GPU_PATTERNS["arrow_big"] = lambda: problems.block_arrow_pattern(12, 64, 128)
GPU_PATTERNS["nested_mid"] = lambda: problems.nested_block_arrow_pattern(nsub=2, nmid=6, nleaf_per_mid=8, seed=3)
GPU_PATTERNS["dense200"] = lambda: problems.band_pattern(200, 199)
GPU_PATTERNS["arrow_thin"] = lambda: problems.block_arrow_pattern(6, 2, 150)   # thin cliques, separators beyond LDS
GPU_PATTERNS["diag"] = lambda: problems.band_pattern(15, 0)          # LP case: every clique is 1 x 1
# single fronts beyond the one-workgroup class (272 rows): the one-launch blocked Cholesky (front_flow.hip) factors the root
# (dense600; arrow_one's 310 x 310 root) and the 300 x 300 Y_AA block of arrow_one's only top front
GPU_PATTERNS["dense600"] = lambda: problems.band_pattern(600, 599)
GPU_PATTERNS["arrow_one"] = lambda: problems.nested_block_arrow_pattern(nsub=1, nmid=2, nleaf_per_mid=2, leaf=(3, 9), mid=(6, 20), top=(40, 300),
                                                                        root=310, seed=9)
# three top fronts in ONE level whose separators (300, 200 and 150 rows of a 310-column root) are beyond the one-workgroup class:
# their chol(Y_AA) runs side by side in one launch of the one-launch blocked Cholesky (front_flow.hip, gridDim.y = 3; orders 5 / 4 / 3
# tiles against a plan laid out for 5)
def _three_tops_pattern():
    root = np.arange(130, 440)
    cl = [(root, root)]
    for s_, na in enumerate((300, 200, 150)):
        own = np.arange(40 * s_, 40 * s_ + 40)
        sep = root[np.sort(np.random.default_rng(90 + s_).choice(len(root), size=na, replace=False))]
        cl.append((own, np.concatenate([own, sep])))
    cl.append((np.arange(120, 130), np.concatenate([np.arange(120, 130), root[:20]])))
    return problems._from_cliques(440, cl)


GPU_PATTERNS["three_tops"] = _three_tops_pattern
# families (front_fam.hip: small parents swept together with their childless children): largest member sizes,
# odd sizes with few children, and nine children per parent (one more than the waves of a workgroup: no family)
GPU_PATTERNS["fam_max"] = lambda: problems.nested_block_arrow_pattern(nsub=1, nmid=3, nleaf_per_mid=8, leaf=(16, 32),
                                                                      mid=(16, 64), top=(40, 50), root=60, seed=5)
GPU_PATTERNS["fam_odd"] = lambda: problems.nested_block_arrow_pattern(nsub=2, nmid=4, nleaf_per_mid=5, leaf=(3, 17),
                                                                      mid=(7, 33), top=(20, 30), root=40, seed=6)
GPU_PATTERNS["fam_nine"] = lambda: problems.nested_block_arrow_pattern(nsub=1, nmid=2, nleaf_per_mid=9, leaf=(2, 9),
                                                                       mid=(6, 20), top=(20, 20), root=30, seed=7)


# odd-sized families under top fronts beyond the LDS class (nf = 130 <= 198): the fused extend-add with three row tiles
GPU_PATTERNS["fam_top"] = lambda: problems.nested_block_arrow_pattern(nsub=2, nmid=5, nleaf_per_mid=3, leaf=(3, 17),
                                                                      mid=(7, 33), top=(30, 100), root=110, seed=8)


def setup(name, seed):
    symb = Symbolic(GPU_PATTERNS[name]())
    symb.device_init(0, 4)
    S = orc.Sym(symb)
    Lh = problems.random_factor_blkval(symb, seed)
    A = Lh.copy()
    orc.llt(S, A)              # S = L L^T on V (oracle used only to manufacture inputs / check)
    return symb, S, A, lowmask(symb)


@pytest.mark.parametrize("name", sorted(GPU_PATTERNS))
def test_cholesky_llt_pinv_completion(name):
    symb, S, A, msk = setup(name, 1)
    X = dev(symb, A)
    chordal.cholesky(X)
    ref = A.copy()
    orc.cholesky(S, ref)
    assert rel(host(X)[msk], ref[msk]) < TOL
    assert abs(chordal.logdiagsum(X) - orc.logdiagsum(S, ref)) < 1e-9 * max(1, abs(orc.logdiagsum(S, ref)))
    Lfac = X.copy()
    # llt round trip
    chordal.llt(X)
    assert rel(host(X)[msk], A[msk]) < TOL
    # projected inverse
    Y = Lfac.copy()
    chordal.projected_inverse(Y)
    yref = ref.copy()
    orc.projected_inverse(S, yref)
    assert rel(host(Y)[msk], yref[msk]) < TOL
    # completion: round trip back to the factor, and against the oracle
    C = Y.copy()
    chordal.completion(C)
    cref = yref.copy()
    orc.completion(S, cref)
    assert rel(host(C)[msk], cref[msk]) < 1e-8
    assert rel(host(C)[msk], ref[msk]) < 1e-8
    # dot
    assert abs(chordal.dot(Y, dev(symb, A)) - orc.dot(S, yref, A)) < 1e-9 * max(1, abs(orc.dot(S, yref, A)))


@pytest.mark.parametrize("name", ["band", "arrow", "rand2"])
def test_failure_is_arithmetic_error(name):
    symb, S, A, msk = setup(name, 2)
    nn, na = symb.clique_sizes()
    k = symb.Nsn // 2
    bad = A.copy()
    bad[symb.blkptr[k]] = -1.0                 # a diagonal entry
    with pytest.raises(ArithmeticError):
        chordal.cholesky(dev(symb, bad))
    with pytest.raises(ArithmeticError):
        chordal.completion(dev(symb, bad))
    # and the context stays usable afterwards
    X = dev(symb, A)
    chordal.cholesky(X)


@pytest.mark.parametrize("name", sorted(GPU_PATTERNS))
@pytest.mark.parametrize("adj,inv", [(None, False), (None, True), (False, False), (True, False),
                                     (False, True), (True, True)])
def test_hessian(name, adj, inv):
    symb, S, A, msk = setup(name, 3)
    rng = np.random.default_rng(4)
    L = A.copy()
    orc.cholesky(S, L)
    Yh = L.copy()
    orc.projected_inverse(S, Yh)
    nr = 4          # >= 4 dense right-hand sides take the family kernel where the pattern has families
    U = rng.standard_normal((nr, symb.blklen)) * msk
    ref = U.copy()
    for r in range(nr):
        orc.hessian(S, L, Yh, ref[r], adj=adj, inv=inv)
    Ud = torch.from_numpy(U).cuda()
    chordal.hessian(dev(symb, L), dev(symb, Yh), Ud, adj=adj, inv=inv)
    got = Ud.cpu().numpy()
    for r in range(nr):
        assert rel(got[r][msk], ref[r][msk]) < 1e-9
    # single-matrix and list forms agree with the batched form
    one = dev(symb, U[0])
    chordal.hessian(dev(symb, L), dev(symb, Yh), [one], adj=adj, inv=inv)
    assert rel(host(one)[msk], ref[0][msk]) < 1e-9


@pytest.mark.parametrize("name", ["band", "arrow", "rand2", "nested", "arrow_big", "dense200", "arrow_thin", "nested_mid", "diag"])
def test_trsm(name):
    symb, S, A, msk = setup(name, 5)
    rng = np.random.default_rng(6)
    L = A.copy()
    orc.cholesky(S, L)
    # 4 right-hand sides: the substitution kernels (front_generic.hip); 8 and 70: tile products with the inverse-form
    # factor (k_trsm_mm_*, front_large.hip; 70 columns = two column tiles, the second one ragged)
    for nrhs in (4, 8, 70):
        for trans in ("N", "T"):
            B = rng.standard_normal((nrhs, symb.n))
            ref = B.copy()
            orc.trsm(S, L, ref, trans)
            Bd = torch.from_numpy(B).cuda()
            chordal.trsm(dev(symb, L), Bd, trans)
            assert rel(Bd.cpu().numpy(), ref) < TOL, (nrhs, trans)


@pytest.mark.parametrize("name", ["arrow", "nested_mid", "fam_odd"])
def test_deferred_status(name):
    """chordal.lazy_status: the factorisations return without reading the device flag back; check_status raises what they
    would have raised, once, and a whole KKT solve under the deferred regime equals the eager one (H, x and y
    to rounding: the extend-adds and Amap sum with atomics)."""
    symb, S, A, msk = setup(name, 2)
    rng = np.random.default_rng(4)
    m = 6
    cptr, cidx, cval = problems.random_constraints(symb, m, density=0.05, seed=9)
    kkt = KKTSystem(symb, cptr, cidx, cval, max_rhs=4)
    b0 = rng.standard_normal(symb.blklen) * msk
    y0 = rng.standard_normal(m)

    def solve(mat):
        L = dev(symb, mat)
        chordal.cholesky(L)
        Y = L.copy()
        chordal.projected_inverse(Y)
        f = kkt.factor(L, Y)
        bx, by = dev(symb, b0), torch.from_numpy(y0.copy()).cuda()
        f(bx, by, 0.7)
        return host(bx), by.cpu().numpy(), kkt.H.cpu().numpy().copy()

    x1, y1, H1 = solve(A)
    bad = A.copy()
    bad[symb.blkptr[symb.Nsn // 2]] = -1.0
    chordal.lazy_status(symb, True)
    try:
        x2, y2, H2 = solve(A)
        chordal.check_status(symb)                       # nothing failed
        assert rel(H2, H1) < 1e-13 and rel(y2, y1) < 1e-12 and rel(x2[msk], x1[msk]) < 1e-12
        chordal.cholesky(dev(symb, bad))                 # returns at once ...
        with pytest.raises(ArithmeticError):
            chordal.check_status(symb)                   # ... the failure is reported here
        chordal.check_status(symb)                       # and only once
        solve(bad)                                       # a whole solve on a matrix outside the cone: no hang, no fault
        with pytest.raises(ArithmeticError):
            chordal.check_status(symb)
        x3, y3, H3 = solve(A)                            # the context recovers
        chordal.check_status(symb)
        assert rel(H3, H1) < 1e-13 and rel(y3, y1) < 1e-12
    finally:
        chordal.lazy_status(symb, False)
    with pytest.raises(ArithmeticError):
        chordal.cholesky(dev(symb, bad))                 # eager again


@pytest.mark.parametrize("name", ["arrow", "rand2", "nested_mid", "diag", "fam_max", "fam_odd", "fam_nine", "nested", "rand1"])
def test_kkt_factor_and_solve(name):
    symb, S, A, msk = setup(name, 7)
    rng = np.random.default_rng(8)
    L = A.copy()
    orc.cholesky(S, L)
    Yh = L.copy()
    orc.projected_inverse(S, Yh)
    m = 7
    cptr, cidx, cval = problems.random_constraints(symb, m, density=0.05, seed=9)
    K = orc.KKT(S, cptr, cidx, cval)
    Href = K.schur_factor(L, Yh)
    sys = KKTSystem(symb, cptr, cidx, cval, max_rhs=3)       # forces chunking over the constraints
    Ld, Yd = dev(symb, L), dev(symb, Yh)
    # Amap / Aadj
    x = rng.standard_normal(symb.blklen) * msk
    assert rel(sys.amap(dev(symb, x)).cpu().numpy(), K.amap(x)) < TOL
    y = rng.standard_normal(m)
    assert rel(host(sys.aadj(torch.from_numpy(y).cuda()))[msk], K.aadj(y)[msk]) < TOL
    solve = sys.factor(Ld, Yd)
    Hg = np.tril(sys.H.cpu().numpy().T)          # device H is column-major m x m
    assert rel(Hg, np.tril(Href)) < 1e-9
    bx = rng.standard_normal(symb.blklen) * msk
    by = rng.standard_normal(m)
    for kk in (1.0, 0.25):
        xr, yr = K.solve(L, Yh, Href, bx, by, kk)
        bxd, byd = dev(symb, bx), torch.from_numpy(by.copy()).cuda()
        solve(bxd, byd, kk)
        assert rel(host(bxd)[msk], xr[msk]) < 1e-9
        assert rel(byd.cpu().numpy(), yr) < 1e-9
        # reference's own DEBUG residual check (solvers.py:534-538) on the GPU result
        r, rr = K.residual(L, Yh, host(bxd) * msk, byd.cpu().numpy(), bx, by, kk)
        assert np.sqrt(orc.dot(S, r, r)) / max(1, np.sqrt(orc.dot(S, bx, bx))) < 1e-10
        assert np.linalg.norm(rr) / max(1, np.linalg.norm(by)) < 1e-10


def _launch_counts(symb, fn):
    """kernel name -> launches while fn() runs (csp_profile_*: HIP events around every launch)"""
    import ctypes
    from smcp_amd import _lib
    lib = _lib.lib()
    h = symb.handle
    nk = int(lib.csp_profile_kinds())
    names = [lib.csp_profile_kernel_name(i).decode() for i in range(nk)]
    lib.csp_profile_filter(h, -1)
    lib.csp_profile_enable(h, 1)
    lib.csp_profile_read(h, None, None)
    try:
        fn()
        torch.cuda.synchronize()
        ms = (ctypes.c_double * nk)()
        cnt = (ctypes.c_int64 * nk)()
        lib.csp_profile_read(h, ms, cnt)
    finally:
        lib.csp_profile_enable(h, 0)
    return {names[i]: int(cnt[i]) for i in range(nk) if cnt[i]}


@pytest.mark.parametrize("name,m,density", [("nested_mid", 12, 0.002), ("fam_odd", 10, 0.03), ("nested", 8, 0.03)])
def test_family_children_closed_form_gram_and_entry_driven_sweep(name, m, density):
    """Round 3: the Schur complement of a tree with families is built WITHOUT the children's panels -- k_fam_terms sweeps
    the parents from the entry lists (front_famt.hip), k_leaf_pairs supplies the children's Gram block in closed form
    (front_leafgram.hip), k_gram_diag128 walks a slice table that leaves the children's rows out.  H, x, y against the
    oracle, and the launch counters show that these kernels are the ones that ran."""
    symb, S, A, msk = setup(name, 11)
    rng = np.random.default_rng(12)
    L = A.copy()
    orc.cholesky(S, L)
    Yh = L.copy()
    orc.projected_inverse(S, Yh)
    cptr, cidx, cval = problems.random_constraints(symb, m, density=density, seed=13)
    K = orc.KKT(S, cptr, cidx, cval)
    Href = K.schur_factor(L, Yh)
    sys = KKTSystem(symb, cptr, cidx, cval, max_rhs=m, tnzcols=0.0)
    chordal.tune(symb, chordal.TUNE_LEAFGRAM, 2)          # the test problems are too small for the cost rule to pick the route
    Ld, Yd = dev(symb, L), dev(symb, Yh)
    box = {}
    counts = _launch_counts(symb, lambda: box.setdefault("solve", sys.factor(Ld, Yd)))
    assert counts.get("k_fam_terms", 0) >= 1 and counts.get("k_leaf_pairs", 0) >= 1 and counts.get("k_leaf_tables", 0) >= 1, counts
    assert counts.get("k_gram_diag128", 0) >= 1, counts
    Hg = np.tril(sys.H.cpu().numpy().T)
    assert rel(Hg, np.tril(Href)) < 1e-9
    bx = rng.standard_normal(symb.blklen) * msk
    by = rng.standard_normal(m)
    xr, yr = K.solve(L, Yh, Href, bx, by, 0.5)
    bxd, byd = dev(symb, bx), torch.from_numpy(by.copy()).cuda()
    box["solve"](bxd, byd, 0.5)
    assert rel(host(bxd)[msk], xr[msk]) < 1e-9 and rel(byd.cpu().numpy(), yr) < 1e-9
    # the QR route needs the whole stack: the children's panels are formed again (k_fam_sparse), same H through Q^T Q
    sysq = KKTSystem(symb, cptr, cidx, cval, max_rhs=m, tnzcols=0.0)
    counts_q = _launch_counts(symb, lambda: sysq.factor_qr(Ld, Yd))
    assert counts_q.get("k_fam_terms", 0) == 0 and counts_q.get("k_leaf_pairs", 0) == 0, counts_q


def test_stale_cache_is_detected_and_touch_repairs_it():
    """the derived-quantity caches are keyed by device address; a C caller that rescales a factor in
    place (the reference's blas.scal(a, X.blkval), solvers.py:407) and forgets csp_touch used to get results from the
    OLD factor.  With CSP_TUNE_VERIFY_CACHE the reuse is refused (SMCP_ESTALE); after csp_touch the call is right."""
    from smcp_amd import _lib
    from smcp_amd.cspmatrix import _stream
    symb, S, A, msk = setup("nested_mid", 3)
    lib = _lib.lib()
    chordal.tune(symb, chordal.TUNE_VERIFY_CACHE, 1)
    try:
        L = dev(symb, A)
        chordal.cholesky(L)
        Y = L.copy()
        chordal.projected_inverse(Y)
        rng = np.random.default_rng(5)
        u0 = rng.standard_normal(symb.blklen) * msk

        def raw_hessian(U):      # the C ABI directly: nothing between the caller's pointers and the library
            return lib.csp_hessian(symb.handle, L.blkval.data_ptr(), Y.blkval.data_ptr(), U.blkval.data_ptr(), 1,
                                   symb.blklen, 2, 0, _stream())

        U1 = dev(symb, u0)
        assert raw_hessian(U1) == 0
        # rescale the scaling point in place behind the library's back: S -> S / 4, i.e. L -> L / 2, Y -> 4 Y
        L.blkval.mul_(0.5)
        Y.blkval.mul_(4.0)
        U2 = dev(symb, u0)
        assert raw_hessian(U2) == -5                      # SMCP_ESTALE: refused, not silently wrong
        assert lib.csp_touch(symb.handle, L.blkval.data_ptr()) == 0 and lib.csp_touch(symb.handle, Y.blkval.data_ptr()) == 0
        U3 = dev(symb, u0)
        assert raw_hessian(U3) == 0
        # H(U) = P_V(S^-1 U S^-1): S / 4 gives 16 times the first result
        assert rel(host(U3)[msk], 16.0 * host(U1)[msk]) < 1e-12
    finally:
        chordal.tune(symb, chordal.TUNE_VERIFY_CACHE, 0)


def test_deterministic_mode_is_bit_identical_from_run_to_run():
    """csp_tune(CSP_TUNE_DETERMINISTIC, 1) routes every tree operation through the fixed-order kernels
    (no floating-point atomics: the default extend-adds sum children with ds_add_f64 in arrival order): H, x and y of
    five whole KKT solves are bit-identical, and agree with the default (atomic) route to rounding."""
    symb, S, A, msk = setup("nested_mid", 21)
    rng = np.random.default_rng(22)
    m = 9
    cptr, cidx, cval = problems.random_constraints(symb, m, density=0.01, seed=23)
    b0 = rng.standard_normal(symb.blklen) * msk
    y0 = rng.standard_normal(m)

    def whole_solve():
        sysk = KKTSystem(symb, cptr, cidx, cval, max_rhs=4, tnzcols=0.0)
        L = dev(symb, A)
        chordal.cholesky(L)
        Y = L.copy()
        chordal.projected_inverse(Y)
        f = sysk.factor(L, Y)
        bx, by = dev(symb, b0), torch.from_numpy(y0.copy()).cuda()
        f(bx, by, 0.7)
        return sysk.H.cpu().numpy().copy(), host(bx).copy(), by.cpu().numpy().copy()

    Hd, xd, yd = whole_solve()                                  # default route
    chordal.tune(symb, chordal.TUNE_DETERMINISTIC, 1)
    try:
        runs = [whole_solve() for _ in range(5)]
    finally:
        chordal.tune(symb, chordal.TUNE_DETERMINISTIC, 0)
    for H, x, y in runs[1:]:
        assert np.array_equal(H, runs[0][0]) and np.array_equal(x, runs[0][1]) and np.array_equal(y, runs[0][2])
    assert rel(runs[0][0], Hd) < 1e-12 and rel(runs[0][1][msk], xd[msk]) < 1e-11 and rel(runs[0][2], yd) < 1e-11


def test_two_kkt_systems_on_one_symbolic_do_not_share_constraints():
    """The constraint set lives in the Symbolic's native context; a KKTSystem re-installs its own set when another
    system has used the context in between (the first system silently ran on the second one's constraints).
    The Q factor of kkt_qr cannot be re-created and is refused instead."""
    symb, S, A, msk = setup("nested_mid", 21)
    rng = np.random.default_rng(22)
    L = A.copy()
    orc.cholesky(S, L)
    Yh = L.copy()
    orc.projected_inverse(S, Yh)
    Ld, Yd = dev(symb, L), dev(symb, Yh)
    cons = [problems.random_constraints(symb, m, density=0.05, seed=sd) for m, sd in ((5, 23), (8, 24))]
    Ks = [orc.KKT(S, *c) for c in cons]
    Hs = [K.schur_factor(L, Yh) for K in Ks]
    sys0 = KKTSystem(symb, *cons[0], max_rhs=4)
    solve0 = sys0.factor(Ld, Yd)
    sys1 = KKTSystem(symb, *cons[1], max_rhs=4)            # takes the context over
    solve1 = sys1.factor(Ld, Yd)
    x = rng.standard_normal(symb.blklen) * msk
    for sy, K, H, solve in ((sys0, Ks[0], Hs[0], solve0), (sys1, Ks[1], Hs[1], solve1), (sys0, Ks[0], Hs[0], solve0)):
        assert rel(sy.amap(dev(symb, x)).cpu().numpy(), K.amap(x)) < TOL
        y = rng.standard_normal(sy.m)
        assert rel(host(sy.aadj(torch.from_numpy(y).cuda()))[msk], K.aadj(y)[msk]) < TOL
        bx, by = rng.standard_normal(symb.blklen) * msk, rng.standard_normal(sy.m)
        xr, yr = K.solve(L, Yh, H, bx, by, 1.0)
        bxd, byd = dev(symb, bx), torch.from_numpy(by.copy()).cuda()
        solve(bxd, byd, 1.0)
        assert rel(host(bxd)[msk], xr[msk]) < 1e-9 and rel(byd.cpu().numpy(), yr) < 1e-9
    q0 = KKTSystem(symb, *cons[0], max_rhs=4, tnzcols=0.0)
    qsolve = q0.factor_qr(Ld, Yd)
    sys1.amap(dev(symb, x))                                # another system uses the context: Q is gone
    with pytest.raises(RuntimeError, match="factor again"):
        qsolve(dev(symb, x), torch.zeros(q0.m, dtype=torch.float64, device="cuda"), 1.0)


@pytest.mark.parametrize("n,pad", [(5, 0), (64, 0), (130, 0), (200, 3), (517, 0), (1000, 0), (1027, 5), (2048, 0), (4096, 0)])
def test_dense_potrf_potrs(n, pad):
    """Schur-complement factor/solve (solvers.py:452,499: lapack.potrf / potrs) at sizes on both sides of the
    switch from the single-workgroup kernel to the blocked one -- since round 5 ONE launch with the tile dependencies
    resolved inside it (front_flow.hip), up to order 4096; a leading dimension beyond the order; a matrix that stops being
    positive definite at a LATE pivot (the failure must reach the caller from inside the launch, and the next call on the
    same context must work)."""
    from smcp_amd import _lib
    symb = Symbolic(GPU_PATTERNS["band"]())
    chordal._ensure(symb)
    rng = np.random.default_rng(n)
    M = rng.standard_normal((n, n))
    Hh = M @ M.T + n * np.eye(n)
    ld = n + pad
    Hp = np.zeros((n, ld))
    Hp[:, :n] = Hh                                       # (row r of this array = column r of the column-major matrix with ld rows)
    H = torch.from_numpy(Hp.copy()).cuda()
    b = rng.standard_normal(n)
    bd = torch.from_numpy(b.copy()).cuda()
    lib = _lib.lib()
    assert lib.dense_potrf(symb.handle, H.data_ptr(), n, ld, None) == 0
    Lref = np.linalg.cholesky(Hh)
    assert rel(np.tril(H.cpu().numpy()[:, :n].T), Lref) < 1e-12
    assert lib.dense_potrs(symb.handle, H.data_ptr(), n, ld, bd.data_ptr(), 1, n, None) == 0
    assert rel(bd.cpu().numpy(), np.linalg.solve(Hh, b)) < 1e-10
    Hbad = torch.from_numpy((Hh - 2 * np.linalg.eigvalsh(Hh)[1] * np.eye(n)).copy()).cuda()
    assert lib.dense_potrf(symb.handle, Hbad.data_ptr(), n, n, None) > 0
    if n > 128:
        # positive definite up to the last 64-column block, not beyond: L L^T with the last pivot's square turned negative
        Hlate = Hh.copy()
        Hlate[n - 1, n - 1] = Lref[n - 1, :n - 1] @ Lref[n - 1, :n - 1] - 1e-3
        Hl = torch.from_numpy(Hlate).cuda()
        assert lib.dense_potrf(symb.handle, Hl.data_ptr(), n, n, None) > 0
        H2 = torch.from_numpy(Hh.copy()).cuda()
        assert lib.dense_potrf(symb.handle, H2.data_ptr(), n, n, None) == 0
        assert rel(np.tril(H2.cpu().numpy().T), Lref) < 1e-12


@pytest.mark.parametrize("name,tnz", [("rand2", 0.3), ("arrow", 0.5), ("nested_mid", 0.2), ("diag", 0.5), ("arrow", 0.0)])
def test_kkt_column_sparse_constraints(name, tnz):
    """Row a9: constraints touching few columns go through the SCMcolumn2 path (solvers.py:489-497,
    misc.c:620-663); the Schur complement and the solve must not depend on which path a constraint took."""
    symb, S, A, msk = setup(name, 11)
    rng = np.random.default_rng(12)
    L = A.copy()
    orc.cholesky(S, L)
    Yh = L.copy()
    orc.projected_inverse(S, Yh)
    valid = problems.lower_positions(symb)
    cidx, cval, cptr = [], [], [0]
    ndiag = problems.lower_positions(symb)
    for j in range(9):
        if j % 3 == 0:      # dense on V
            idx = valid
        elif j % 3 == 1:    # a handful of entries
            idx = np.sort(rng.choice(valid, size=min(3, len(valid)), replace=False))
        else:               # a single entry
            idx = np.sort(rng.choice(valid, size=1))
        cidx.append(idx)
        cval.append(rng.standard_normal(len(idx)))
        cptr.append(cptr[-1] + len(idx))
    cptr = np.asarray(cptr, dtype=np.int64)
    cidx = np.concatenate(cidx).astype(np.int64)
    cval = np.concatenate(cval)
    m = len(cptr) - 1
    K = orc.KKT(S, cptr, cidx, cval)
    Href = K.schur_factor(L, Yh)
    sys = KKTSystem(symb, cptr, cidx, cval, max_rhs=4, tnzcols=tnz)
    Ld, Yd = dev(symb, L), dev(symb, Yh)
    solve = sys.factor(Ld, Yd)
    Hg = np.tril(sys.H.cpu().numpy().T)
    assert rel(Hg, np.tril(Href)) < 1e-9
    bx = rng.standard_normal(symb.blklen) * msk
    by = rng.standard_normal(m)
    xr, yr = K.solve(L, Yh, Href, bx, by, 1.0)
    bxd, byd = dev(symb, bx), torch.from_numpy(by.copy()).cuda()
    solve(bxd, byd, 1.0)
    assert rel(host(bxd)[msk], xr[msk]) < 1e-8
    assert rel(byd.cpu().numpy(), yr) < 1e-8


def test_kkt_many_constraints():
    """m > 128: the Gram matrix spans several 128-column blocks (diagonal and off-diagonal block kernels) and
    H takes the blocked dense Cholesky / blocked triangular solves."""
    symb, S, A, msk = setup("nested_mid", 21)
    rng = np.random.default_rng(22)
    L = A.copy()
    orc.cholesky(S, L)
    Yh = L.copy()
    orc.projected_inverse(S, Yh)
    m = 150
    cptr, cidx, cval = problems.random_constraints(symb, m, density=0.02, seed=23)
    K = orc.KKT(S, cptr, cidx, cval)
    Href = K.schur_factor(L, Yh)
    sys = KKTSystem(symb, cptr, cidx, cval, max_rhs=64, tnzcols=0.0)
    Ld, Yd = dev(symb, L), dev(symb, Yh)
    solve = sys.factor(Ld, Yd)
    Hg = np.tril(sys.H.cpu().numpy().T)
    assert rel(Hg, np.tril(Href)) < 1e-9
    bx = rng.standard_normal(symb.blklen) * msk
    by = rng.standard_normal(m)
    xr, yr = K.solve(L, Yh, Href, bx, by, 0.5)
    bxd, byd = dev(symb, bx), torch.from_numpy(by.copy()).cuda()
    solve(bxd, byd, 0.5)
    assert rel(host(bxd)[msk], xr[msk]) < 1e-8
    assert rel(byd.cpu().numpy(), yr) < 1e-8


def test_kkt_wide_separators_stream_through_lds_front():
    """Extend-add of a large front whose packed lower triangle fits LDS (k_lf_assemble_lds) with child separators
    beyond 128 rows (the per-lane row registers cover 128; the rest goes through the plain loop): thin cliques (2, 150)
    under a (150, 0) root, 40 constraints in one chunk so that the streaming kernel is selected."""
    symb = Symbolic(problems.block_arrow_pattern(6, 2, 150))
    symb.device_init(0, 40)
    S = orc.Sym(symb)
    A = problems.random_factor_blkval(symb, 31)
    orc.llt(S, A)
    L = A.copy()
    orc.cholesky(S, L)
    Yh = L.copy()
    orc.projected_inverse(S, Yh)
    m = 40
    cptr, cidx, cval = problems.random_constraints(symb, m, density=0.03, seed=32)
    K = orc.KKT(S, cptr, cidx, cval)
    Href = K.schur_factor(L, Yh)
    sys_ = KKTSystem(symb, cptr, cidx, cval, max_rhs=40, tnzcols=0.0)
    sys_.factor(dev(symb, L), dev(symb, Yh))
    assert rel(np.tril(sys_.H.cpu().numpy().T), np.tril(Href)) < 1e-9
    # dense multi-RHS Hessian through the same level (40 right-hand sides)
    rng = np.random.default_rng(33)
    msk = lowmask(symb)
    U = rng.standard_normal((40, symb.blklen)) * msk
    ref = U.copy()
    for r in range(0, 40, 13):
        orc.hessian(S, L, Yh, ref[r], adj=None, inv=False)
    Ud = torch.from_numpy(U).cuda()
    chordal.hessian(dev(symb, L), dev(symb, Yh), Ud, adj=None, inv=False)
    got = Ud.cpu().numpy()
    for r in range(0, 40, 13):
        assert rel(got[r][msk], ref[r][msk]) < 1e-9


def test_kkt_wide_dense_clique_sparse_first_phase():
    """Config-2 shape at test size: ONE dense clique of 450 columns (wide enough for the symmetric split of the large-front
    sweep, seven column tiles with a ragged last one) and sparse constraints -- the first phase of the Schur sweep is
    k_lf_zsp (Z = Li Fl as a sparse combination of columns of Li), no dense panel is built; H against the oracle and
    against the dense route (SMCP_ZSP has no run-time switch: the launch counts tell which route ran)."""
    symb = Symbolic(problems.band_pattern(450, 449))
    symb.device_init(0, 5)
    S = orc.Sym(symb)
    A = problems.random_factor_blkval(symb, 41)
    orc.llt(S, A)
    L = A.copy()
    orc.cholesky(S, L)
    Yh = L.copy()
    orc.projected_inverse(S, Yh)
    m = 5
    cptr, cidx, cval = problems.random_constraints(symb, m, density=0.01, seed=42)
    K = orc.KKT(S, cptr, cidx, cval)
    Href = K.schur_factor(L, Yh)
    sys_ = KKTSystem(symb, cptr, cidx, cval, max_rhs=3, tnzcols=0.0)
    Ld, Yd = dev(symb, L), dev(symb, Yh)
    counts = _launch_counts(symb, lambda: sys_.factor(Ld, Yd))
    assert counts.get("k_lf_zsp", 0) >= 1 and counts.get("k_lf_up1", 0) == 0, counts
    assert rel(np.tril(sys_.H.cpu().numpy().T), np.tril(Href)) < 1e-9
    rng = np.random.default_rng(43)
    msk = lowmask(symb)
    bx, by = rng.standard_normal(symb.blklen) * msk, rng.standard_normal(m)
    xr, yr = K.solve(L, Yh, Href, bx, by, 1.0)
    bxd, byd = dev(symb, bx), torch.from_numpy(by.copy()).cuda()
    sys_.factor(Ld, Yd)(bxd, byd, 1.0)
    assert rel(host(bxd)[msk], xr[msk]) < 1e-9 and rel(byd.cpu().numpy(), yr) < 1e-9


@pytest.mark.parametrize("nleaf", [12, 9, 3])
def test_kkt_sibling_groups_send_one_summed_update(nleaf, monkeypatch):
    """Config-3 shape at test size: `nleaf` childless (64, 128) fronts under one (128, 0) root, all with the root's 128
    columns as separator.  The sparse-input sweep sums the updates of up to eight siblings in one workgroup and stores ONE
    packed update per group (k_lfsp_up<..., GRP>); the root's extend-add skips the other members.  12 leaves = groups of
    8 + 4, 9 = 8 + 1 (a singleton group), 3 = one group of three.  H and the solve against the oracle, and H against the ungrouped route
    (SMCP_LFSP_GROUP=0 at device_init) to rounding."""
    pat = problems.block_arrow_pattern(nleaf, 64, 128)
    m = 6
    Hs = []
    for grouped in (True, False):
        if not grouped:
            monkeypatch.setenv("SMCP_LFSP_GROUP", "0")
        symb = Symbolic(pat)
        symb.device_init(0, m)
        S = orc.Sym(symb)
        A = problems.random_factor_blkval(symb, 51)
        orc.llt(S, A)
        L = A.copy()
        orc.cholesky(S, L)
        Yh = L.copy()
        orc.projected_inverse(S, Yh)
        cptr, cidx, cval = problems.random_constraints(symb, m, density=0.004, seed=52)
        sys_ = KKTSystem(symb, cptr, cidx, cval, max_rhs=m, tnzcols=0.0)
        Ld, Yd = dev(symb, L), dev(symb, Yh)
        counts = _launch_counts(symb, lambda: sys_.factor(Ld, Yd))
        assert counts.get("k_lfsp_up", 0) == 3, counts
        Hs.append(np.tril(sys_.H.cpu().numpy().T))
        if grouped:
            K = orc.KKT(S, cptr, cidx, cval)
            Href = K.schur_factor(L, Yh)
            assert rel(Hs[0], np.tril(Href)) < 1e-9
            rng = np.random.default_rng(53)
            msk = lowmask(symb)
            bx, by = rng.standard_normal(symb.blklen) * msk, rng.standard_normal(m)
            xr, yr = K.solve(L, Yh, Href, bx, by, 1.0)
            bxd, byd = dev(symb, bx), torch.from_numpy(by.copy()).cuda()
            sys_.factor(Ld, Yd)(bxd, byd, 1.0)
            assert rel(host(bxd)[msk], xr[msk]) < 1e-9 and rel(byd.cpu().numpy(), yr) < 1e-9
    assert rel(Hs[0], Hs[1]) < 1e-12


@pytest.mark.parametrize("nmid,nleaf,m", [(19, 3, 12), (9, 3, 13), (8, 8, 21)])
def test_kkt_family_sibling_groups_send_one_summed_update(nmid, nleaf, m, monkeypatch):
    """synth50k's shape with ONE separator per subtree: `nmid` family parents (each with `nleaf` childless children) under one
    large front, all with that front's own 64 columns as separator (synth50k itself draws every separator at random: no two
    siblings share one, and its family sweep stays k_fam_terms).  The entry-driven family sweep takes the parents by sibling groups of up to
    eight (k_fam_terms_grp): the members' tables pass through LDS in turn, the update tiles of a right-hand side are summed in
    the accumulators and ONE packed update per group reaches the exchange buffer; the extend-add above skips the other
    members.  19 parents = groups of 8 + 8 + 3, 9 = 8 + a singleton, 8 = one full group; 12 / 13 / 21 constraints = one full
    and one ragged round of right-hand sides per workgroup column.  H and the solve against the oracle, and H against the
    ungrouped route (SMCP_FAMT_GROUP=0 at device_init) to rounding."""
    pat = problems.nested_block_arrow_pattern(nsub=1, nmid=nmid, nleaf_per_mid=nleaf, seed=71, shared_mid_sep=True)
    Hs = []
    for grouped in (True, False):
        if not grouped:
            monkeypatch.setenv("SMCP_FAMT_GROUP", "0")
        symb = Symbolic(pat)
        symb.device_init(0, m)
        S = orc.Sym(symb)
        A = problems.random_factor_blkval(symb, 72)
        orc.llt(S, A)
        L = A.copy()
        orc.cholesky(S, L)
        Yh = L.copy()
        orc.projected_inverse(S, Yh)
        cptr, cidx, cval = problems.random_constraints(symb, m, density=0.003, seed=73)
        sys_ = KKTSystem(symb, cptr, cidx, cval, max_rhs=m, tnzcols=0.0)
        Ld, Yd = dev(symb, L), dev(symb, Yh)
        counts = _launch_counts(symb, lambda: sys_.factor(Ld, Yd))
        if grouped:
            assert counts.get("k_fam_terms_grp", 0) >= 1 and counts.get("k_fam_terms", 0) == 0, counts
        else:
            assert counts.get("k_fam_terms_grp", 0) == 0 and counts.get("k_fam_terms", 0) >= 1, counts
        Hs.append(np.tril(sys_.H.cpu().numpy().T))
        if grouped:
            K = orc.KKT(S, cptr, cidx, cval)
            Href = K.schur_factor(L, Yh)
            assert rel(Hs[0], np.tril(Href)) < 1e-9
            rng = np.random.default_rng(74)
            msk = lowmask(symb)
            bx, by = rng.standard_normal(symb.blklen) * msk, rng.standard_normal(m)
            xr, yr = K.solve(L, Yh, Href, bx, by, 1.0)
            bxd, byd = dev(symb, bx), torch.from_numpy(by.copy()).cuda()
            sys_.factor(Ld, Yd)(bxd, byd, 1.0)
            assert rel(host(bxd)[msk], xr[msk]) < 1e-9 and rel(byd.cpu().numpy(), yr) < 1e-9
    assert rel(Hs[0], Hs[1]) < 1e-12


def test_kkt_few_fronts_many_children_deal_children_over_workgroups():
    """One (64, 128) front with 40 children and 20 constraints in one chunk: fewer (front, right-hand side) pairs than half
    the CUs, so the streaming extend-add deals the children of a pair over several workgroups whose partial fronts meet in
    global memory by atomics (the share of ONE rank of an eight-rank job on synth50k; lf_assemble's second nz rule).  H and
    the solve against the oracle."""
    symb = Symbolic(problems.nested_block_arrow_pattern(nsub=1, nmid=40, nleaf_per_mid=2, seed=11))
    m = 20
    symb.device_init(0, m)
    S = orc.Sym(symb)
    A = problems.random_factor_blkval(symb, 61)
    orc.llt(S, A)
    L = A.copy()
    orc.cholesky(S, L)
    Yh = L.copy()
    orc.projected_inverse(S, Yh)
    cptr, cidx, cval = problems.random_constraints(symb, m, density=0.01, seed=62)
    K = orc.KKT(S, cptr, cidx, cval)
    Href = K.schur_factor(L, Yh)
    sys_ = KKTSystem(symb, cptr, cidx, cval, max_rhs=m, tnzcols=0.0)
    Ld, Yd = dev(symb, L), dev(symb, Yh)
    counts = _launch_counts(symb, lambda: sys_.factor(Ld, Yd))
    assert counts.get("k_lf_assemble_lds", 0) + counts.get("k_lf_assemble_lds_dyn", 0) >= 1, counts
    assert counts.get("k_lf_clear_upd", 0) >= 1, counts          # the partial fronts are added into cleared blocks
    assert rel(np.tril(sys_.H.cpu().numpy().T), np.tril(Href)) < 1e-9
    rng = np.random.default_rng(63)
    msk = lowmask(symb)
    bx, by = rng.standard_normal(symb.blklen) * msk, rng.standard_normal(m)
    xr, yr = K.solve(L, Yh, Href, bx, by, 1.0)
    bxd, byd = dev(symb, bx), torch.from_numpy(by.copy()).cuda()
    sys_.factor(Ld, Yd)(bxd, byd, 1.0)
    assert rel(host(bxd)[msk], xr[msk]) < 1e-9 and rel(byd.cpu().numpy(), yr) < 1e-9


def test_placement_tuning_moves_the_exchange_buffer_and_changes_nothing_else():
    """csp_tune(CSP_TUNE_PLACEMENT): the packed exchange buffer is re-allocated a few times for the store-pattern probe; the
    Schur complement and the solve after it are what they were before (every kernel takes the buffer from the context)."""
    import ctypes
    from smcp_amd import _lib
    symb = Symbolic(problems.nested_block_arrow_pattern(nsub=2, nmid=40, nleaf_per_mid=3, seed=12))      # 80 family parents
    m = 12
    symb.device_init(0, m)
    S = orc.Sym(symb)
    A = problems.random_factor_blkval(symb, 71)
    orc.llt(S, A)
    L = A.copy()
    orc.cholesky(S, L)
    Yh = L.copy()
    orc.projected_inverse(S, Yh)
    cptr, cidx, cval = problems.random_constraints(symb, m, density=0.01, seed=72)
    sys_ = KKTSystem(symb, cptr, cidx, cval, max_rhs=m, tnzcols=0.0)
    Ld, Yd = dev(symb, L), dev(symb, Yh)
    sys_.factor(Ld, Yd)
    H0 = sys_.H.clone()
    chordal.tune(symb, chordal.TUNE_PLACEMENT, 3)
    rep = (ctypes.c_double * 3)()
    assert _lib.lib().csp_tune_report(symb.handle, rep) == 0
    assert rep[0] > 0.0 and 0.0 < rep[1] <= rep[0]
    solve = sys_.factor(Ld, Yd)
    assert rel(np.tril(sys_.H.cpu().numpy().T), np.tril(H0.cpu().numpy().T)) < 1e-13
    K = orc.KKT(S, cptr, cidx, cval)
    Href = K.schur_factor(L, Yh)
    rng = np.random.default_rng(73)
    msk = lowmask(symb)
    bx, by = rng.standard_normal(symb.blklen) * msk, rng.standard_normal(m)
    xr, yr = K.solve(L, Yh, Href, bx, by, 1.0)
    bxd, byd = dev(symb, bx), torch.from_numpy(by.copy()).cuda()
    solve(bxd, byd, 1.0)
    assert rel(host(bxd)[msk], xr[msk]) < 1e-9 and rel(byd.cpu().numpy(), yr) < 1e-9


def test_kkt_family_kernel_dense_constraints():
    """Family kernel with long entry lists: constraints dense on V give a (5,31) child 180 entries (more than the 64
    prefetched per wave) and a (15,64) parent 1185 (more than the 256 prefetched per group): the direct-load tails
    of the entry loops."""
    symb, S, A, msk = setup("nested_mid", 41)
    L = A.copy()
    orc.cholesky(S, L)
    Yh = L.copy()
    orc.projected_inverse(S, Yh)
    m = 4
    cptr, cidx, cval = problems.random_constraints(symb, m, density=1.0, seed=42)
    assert (np.diff(cptr) > 0.9 * msk.sum()).all()
    K = orc.KKT(S, cptr, cidx, cval)
    Href = K.schur_factor(L, Yh)
    sys_ = KKTSystem(symb, cptr, cidx, cval, max_rhs=4, tnzcols=0.0)
    sys_.factor(dev(symb, L), dev(symb, Yh))
    assert rel(np.tril(sys_.H.cpu().numpy().T), np.tril(Href)) < 1e-9


@pytest.mark.parametrize("spread,tol", [(0, 1e-12), (3, 1e-9), (5, 1e-5)])
def test_kkt_solve_at_ill_conditioned_scaling_points(spread, tol):
    """Late interior-point iterations scale by matrices whose pivots spread over many decades.  The HIP path (explicit
    inverses of the diagonal blocks, GEMM-only sweeps) must degrade no faster than the oracle (triangular solves): the
    search directions agree to rounding x condition, and the residuals of the defining equations (solvers.py:401-411),
    evaluated by the oracle for both, are of the same size."""
    symb = Symbolic(problems.nested_block_arrow_pattern(nsub=2, nmid=6, nleaf_per_mid=8, seed=3))
    symb.device_init(0, 8)
    S = orc.Sym(symb)
    msk = lowmask(symb)
    rng = np.random.default_rng(10 + spread)
    Lh = problems.random_factor_blkval(symb, 5)
    nn, na = symb.clique_sizes()
    for k in range(symb.Nsn):
        nf = nn[k] + na[k]
        blk = Lh[symb.blkptr[k]:symb.blkptr[k] + nf * nn[k]].reshape((nf, nn[k]), order="F")
        blk *= (10.0 ** (-spread * rng.random(nn[k])))[None, :]
    A = Lh.copy()
    orc.llt(S, A)
    L = A.copy()
    orc.cholesky(S, L)
    Yh = L.copy()
    orc.projected_inverse(S, Yh)
    m = 8
    cptr, cidx, cval = problems.random_constraints(symb, m, density=0.05, seed=9)
    K = orc.KKT(S, cptr, cidx, cval)
    Href = K.schur_factor(L, Yh)
    sys_ = KKTSystem(symb, cptr, cidx, cval, max_rhs=8)
    Ld = dev(symb, A)
    chordal.cholesky(Ld)
    Yd = Ld.copy()
    chordal.projected_inverse(Yd)
    solve = sys_.factor(Ld, Yd)
    bx = rng.standard_normal(symb.blklen) * msk
    by = rng.standard_normal(m)
    xr, yr = K.solve(L, Yh, Href, bx, by, 1.0)
    bxd, byd = dev(symb, bx), torch.from_numpy(by.copy()).cuda()
    solve(bxd, byd, 1.0)
    xg, yg = host(bxd), byd.cpu().numpy()
    assert np.linalg.norm((xg - xr)[msk]) / np.linalg.norm(xr[msk]) < tol
    assert np.linalg.norm(yg - yr) / np.linalg.norm(yr) < tol

    def res(x, y):
        r, rr = K.residual(L, Yh, x * msk, y, bx, by, 1.0)
        return np.sqrt(orc.dot(S, r, r)), np.linalg.norm(rr)

    (g1, g2), (o1, o2) = res(xg, yg), res(xr, yr)
    assert g1 <= 10 * o1 + 1e-12 and g2 <= 10 * o2 + 1e-10


@pytest.mark.parametrize("name", ["nested_mid", "arrow_big", "band", "fam_odd"])
def test_concurrent_cone_probes_match_sequential_factorisations(name):
    """Device-resident line search (csp_probe_*): eight trial factorisations on eight streams with private workspace
    slots give the same in-cone / out-of-cone answers as eight factorisations one after the other, for both cones
    (cholesky: K_V, completion: C_V), on LDS-class and large fronts; the main context's caches are untouched."""
    symb, S, A, msk = setup(name, 51)
    symb.device_init(0, 8)
    X = dev(symb, A)
    Dm = dev(symb, problems.random_factor_blkval(symb, 52))
    chordal.llt(Dm)
    Dm *= -1.0
    als = [0.02 * 2 ** k for k in range(8)]
    # a cached (L, Y) pair of the main context must survive the probes
    Lc = X.copy()
    chordal.cholesky(Lc)
    Yc = Lc.copy()
    chordal.projected_inverse(Yc)
    U0 = np.random.default_rng(53).standard_normal(symb.blklen) * msk
    Ua = dev(symb, U0)
    chordal.hessian(Lc, Yc, Ua, adj=None, inv=False)
    for kind, op in (("d", chordal.cholesky), ("p", chordal.completion)):
        got = chordal.probe_cone(X, Dm, als, kind)
        want = []
        for al in als:
            T = X + Dm * al
            try:
                op(T)
                want.append(True)
            except ArithmeticError:
                want.append(False)
        assert got == want and want[0] and not want[-1]
    Ub = dev(symb, U0)
    chordal.hessian(Lc, Yc, Ub, adj=None, inv=False)
    assert rel(host(Ub)[msk], host(Ua)[msk]) < 1e-13


@pytest.mark.parametrize("name", ["nested_mid", "arrow_big", "band", "fam_odd", "rand2", "diag"])
def test_trial_factorisations_against_the_oracle(name):
    """Step-length probes on the replicated pattern (csp_symbolic_replicate + one csp_cholesky / csp_completion for all
    trials): the in-cone verdict of every trial equals the ORACLE's cholesky / completion of the same trial matrix,
    and the factors of the trials inside the cone match the oracle's factors."""
    symb, S, A, msk = setup(name, 71)
    D = problems.random_factor_blkval(symb, 72)
    orc.llt(S, D)
    D *= -1.0
    X, Dm = dev(symb, A), dev(symb, D)
    for K, als in ((8, [0.02 * 2 ** k for k in range(8)]), (5, [0.0, 0.3, 0.01, 5.0, 0.05])):
        for kind, op in (("d", orc.cholesky), ("p", orc.completion)):
            ok, fac = chordal.probe_factors(X, Dm, als, kind)
            want = []
            for k, al in enumerate(als):
                T = A + al * D
                try:
                    op(S, T)
                    want.append(True)
                    assert ok[k] and rel(host(fac[k])[msk], T[msk]) < TOL
                except ArithmeticError:
                    want.append(False)
            assert ok == want and any(want) and not all(want)
    # the base context is untouched: an ordinary factorisation still works and gives the oracle's factor
    Lc = X.copy()
    chordal.cholesky(Lc)
    Lo = A.copy()
    orc.cholesky(S, Lo)
    assert rel(host(Lc)[msk], Lo[msk]) < TOL


@pytest.mark.parametrize("name", ["arrow_big", "nested_mid"])
def test_failed_factorisation_does_not_poison_later_calls(name):
    """A trial matrix outside the cone (ArithmeticError from completion / cholesky, the everyday event of a line search)
    must not leave the device failure flag behind: kernels of later calls return early on a set flag, and kkt_qr's
    solve_ -- which refactors Y_AA without a failure check of its own -- then worked with a stale factor (found as a
    36 -> 42 iteration drift of the arrow_feas_qr golden case once a line search probed on the main context)."""
    symb, S, msk, L, Yh, cptr, cidx, cval = _kkt_qr_case(name, 8, 81)
    sys_ = KKTSystem(symb, cptr, cidx, cval, max_rhs=8, tnzcols=0.0)
    Ld, Yd = dev(symb, L), dev(symb, Yh)
    rng = np.random.default_rng(82)
    b0 = rng.standard_normal(symb.blklen) * msk
    y0 = rng.standard_normal(8)
    bad = S.project(np.eye(symb.n)) * -1.0                     # negative definite: every factorisation of it fails
    for factor in (sys_.factor_qr, sys_.factor):
        solve = factor(Ld, Yd)

        def run():
            bx, by = dev(symb, b0), torch.from_numpy(y0.copy()).cuda()
            solve(bx, by, 0.7)
            return host(bx), by.cpu().numpy()

        x1, y1 = run()
        for op in (chordal.completion, chordal.cholesky):
            with pytest.raises(ArithmeticError):
                op(dev(symb, bad))
            x2, y2 = run()
            assert rel(x2[msk], x1[msk]) < 1e-12 and rel(y2, y1) < 1e-12


def _kkt_qr_case(name, m, seed, density=0.05):
    symb, S, A, msk = setup(name, seed)
    L = A.copy()
    orc.cholesky(S, L)
    Yh = L.copy()
    orc.projected_inverse(S, Yh)
    cptr, cidx, cval = problems.random_constraints(symb, m, density=density, seed=seed + 2)
    return symb, S, msk, L, Yh, cptr, cidx, cval


@pytest.mark.parametrize("name", ["arrow", "rand2", "nested_mid", "fam_max", "fam_nine", "nested", "band"])
def test_kkt_qr_factor_and_solve(name):
    """kkt_qr (solvers.py:413-475) on the device against its restatement around the oracle (Householder QR of the
    same stack): R up to the svec normalisation, Q orthonormal, (x, y) and the reference's DEBUG residuals."""
    m = 7
    symb, S, msk, L, Yh, cptr, cidx, cval = _kkt_qr_case(name, m, 21)
    rng = np.random.default_rng(22)
    K = orc.KKT(S, cptr, cidx, cval)
    F = K.qr_factor(L, Yh)
    sys = KKTSystem(symb, cptr, cidx, cval, max_rhs=3, tnzcols=0.0)
    Ld, Yd = dev(symb, L), dev(symb, Yh)
    solve = sys.factor_qr(Ld, Yd)
    assert sys.qr_passes == 2 and sys.qr_shift == 0.0
    Rt, G = sys.qr_inspect()
    assert np.abs(G.cpu().numpy() - np.eye(m)).max() < 1e-13
    # R^T R = At^T At: the device's inner product carries the weights, the reference's svec scaling halves it
    assert rel(np.tril(Rt) @ np.tril(Rt).T, 2.0 * F["R"].T @ F["R"]) < 1e-11
    # R itself (positive diagonal on both sides after fixing the signs of the Householder factor)
    Rref = F["R"] * np.sign(np.diag(F["R"]))[:, None] * np.sqrt(2.0)
    assert rel(np.tril(Rt).T, Rref) < 1e-9
    bx = rng.standard_normal(symb.blklen) * msk
    by = rng.standard_normal(m)
    for kk in (1.0, 0.25):
        xr, yr = K.qr_solve(L, Yh, F, bx, by, kk)
        bxd, byd = dev(symb, bx), torch.from_numpy(by.copy()).cuda()
        # another factorisation between factor and solve (what a line search does) must not disturb the closure: its
        # half-Hessians need chol(Y_AA) of THIS Y, which the completion below replaces in the cache
        other = dev(symb, Yh * (1.0 + 0.5 * kk))
        chordal.completion(other)
        solve(bxd, byd, kk)
        assert rel(host(bxd)[msk], xr[msk]) < 1e-9
        assert rel(byd.cpu().numpy(), yr) < 1e-9
        r, rr = K.residual(L, Yh, host(bxd) * msk, byd.cpu().numpy(), bx, by, kk)
        assert np.sqrt(orc.dot(S, r, r)) / max(1, np.sqrt(orc.dot(S, bx, bx))) < 1e-10
        assert np.linalg.norm(rr) / max(1, np.linalg.norm(by)) < 1e-10
    # the chol-based solver on the same system afterwards (it rewrites the stack): the QR closure must refuse
    sys.factor(Ld, Yd)
    with pytest.raises(Exception):
        solve(dev(symb, bx), torch.from_numpy(by.copy()).cuda(), 1.0)


def test_kkt_qr_many_constraints():
    """m = 150 constraints (more than one 128-column Gram block, 19 accumulator blocks in k_stack_rmul)."""
    m = 150
    symb, S, msk, L, Yh, cptr, cidx, cval = _kkt_qr_case("nested_mid", m, 31, density=0.02)
    rng = np.random.default_rng(32)
    K = orc.KKT(S, cptr, cidx, cval)
    F = K.qr_factor(L, Yh)
    sys = KKTSystem(symb, cptr, cidx, cval, max_rhs=16, tnzcols=0.0)
    solve = sys.factor_qr(dev(symb, L), dev(symb, Yh))
    Rt, G = sys.qr_inspect()
    assert np.abs(G.cpu().numpy() - np.eye(m)).max() < 1e-12
    bx = rng.standard_normal(symb.blklen) * msk
    by = rng.standard_normal(m)
    xr, yr = K.qr_solve(L, Yh, F, bx, by, 0.5)
    bxd, byd = dev(symb, bx), torch.from_numpy(by.copy()).cuda()
    solve(bxd, byd, 0.5)
    assert rel(host(bxd)[msk], xr[msk]) < 1e-8
    assert rel(byd.cpu().numpy(), yr) < 1e-8


def test_kkt_qr_more_than_320_constraints():
    """the triangular solve of the stack kept a row block per wave in registers (m <= 320); wider
    factors now go through panels of 256 rows -- the rows above a panel are eliminated by a tall product on the matrix
    pipe (k_stack_gemm_panel), the substitution runs inside the panel.  m = 420: two panels, the second one ragged."""
    m = 420
    symb, S, msk, L, Yh, cptr, cidx, cval = _kkt_qr_case("nested_mid", m, 51, density=0.004)
    rng = np.random.default_rng(52)
    K = orc.KKT(S, cptr, cidx, cval)
    F = K.qr_factor(L, Yh)
    sys = KKTSystem(symb, cptr, cidx, cval, max_rhs=32, tnzcols=0.0)
    solve = sys.factor_qr(dev(symb, L), dev(symb, Yh))
    Rt, G = sys.qr_inspect()
    assert np.abs(G.cpu().numpy() - np.eye(m)).max() < 1e-11
    assert rel(np.tril(Rt) @ np.tril(Rt).T, 2.0 * F["R"].T @ F["R"]) < 1e-10
    bx = rng.standard_normal(symb.blklen) * msk
    by = rng.standard_normal(m)
    xr, yr = K.qr_solve(L, Yh, F, bx, by, 0.5)
    bxd, byd = dev(symb, bx), torch.from_numpy(by.copy()).cuda()
    solve(bxd, byd, 0.5)
    assert rel(host(bxd)[msk], xr[msk]) < 1e-8
    assert rel(byd.cpu().numpy(), yr) < 1e-8


@pytest.mark.parametrize("eps,shifted,lazy", [(1e-9, False, False), (1e-13, True, False), (1e-13, True, True)])
def test_kkt_qr_nearly_dependent_constraints(eps, shifted, lazy):
    """Two constraints that differ by eps times a third, independent one: kappa(At) ~ 2 / eps.  At 1e-9 chol(At^T At)
    still goes through and a third pass is planned from the deviation of the second Gram matrix; at 1e-13 it breaks
    down and the first pass is repeated on the shifted Gram matrix (shifted CholeskyQR3).  Either way the QR path must
    deliver the residuals of the reference's DEBUG check; the Householder restatement is the yardstick.
    lazy: with chordal.lazy_status on (what bench.py runs) the breakdown must still be SEEN by kkt_qr_factor -- the flag
    reads that choose between the plain and the shifted route are eager whatever the mode."""
    m = 6
    symb, S, msk, L, Yh, cptr, cidx, cval = _kkt_qr_case("nested_mid", m + 1, 41, density=0.05)
    K0 = orc.KKT(S, cptr, cidx, cval)
    dense = np.stack([K0.constraint(j) for j in range(m + 1)])
    dense[1] = dense[0] + eps * dense[m]         # the perturbation is independent of the other constraints
    cptr2, cidx2, cval2 = [0], [], []
    for j in range(m):
        nz = np.flatnonzero(dense[j])
        cidx2.extend(nz.tolist()); cval2.extend(dense[j][nz].tolist()); cptr2.append(len(cidx2))
    cptr2, cidx2, cval2 = np.array(cptr2), np.array(cidx2), np.array(cval2)
    K = orc.KKT(S, cptr2, cidx2, cval2)
    F = K.qr_factor(L, Yh)
    assert np.linalg.cond(F["R"]) > 1e8
    sys = KKTSystem(symb, cptr2, cidx2, cval2, max_rhs=4, tnzcols=0.0)
    if lazy:
        chordal.lazy_status(symb, True)
    try:
        solve = sys.factor_qr(dev(symb, L), dev(symb, Yh))
        if lazy:
            chordal.check_status(symb)               # nothing latched: the shifted route succeeded
    finally:
        if lazy:
            chordal.lazy_status(symb, False)
    Rt, G = sys.qr_inspect()
    assert np.abs(G.cpu().numpy() - np.eye(m)).max() < 1e-10
    rng = np.random.default_rng(42)
    bx = rng.standard_normal(symb.blklen) * msk
    by = rng.standard_normal(m) * 0.0          # consistent right-hand side: A x = 0
    bxd, byd = dev(symb, bx), torch.from_numpy(by.copy()).cuda()
    solve(bxd, byd, 1.0)
    x, y = host(bxd) * msk, byd.cpu().numpy()
    # kappa(At^T At) ~ 4e18 at eps = 1e-9: whether its Cholesky goes through is decided by the last bits of the Gram
    # matrix (the order of its partial sums), so only the breakdown at 1e-13 is asserted; three passes either way
    assert sys.qr_passes == 3 and (sys.qr_shift > 0.0 or not shifted)
    res = lambda xx, yy: (np.sqrt(orc.dot(S, *(2 * [K.residual(L, Yh, xx, yy, bx, by, 1.0)[0]]))) / max(1, np.sqrt(orc.dot(S, bx, bx))),
                          np.linalg.norm(K.residual(L, Yh, xx, yy, bx, by, 1.0)[1]))
    xr, yr = K.qr_solve(L, Yh, F, bx, by, 1.0)
    (r1, r2), (o1, o2) = res(x, y), res(xr, yr)
    # the achievable residual grows with |y| ~ kappa; the Householder solution of the same system is the measure
    assert r1 < max(1e-8, 3 * o1) and r2 < 1e-10
    assert rel(x[msk], xr[msk]) < max(1e-7, 20 * o1)     # x is well determined even though y is not


def test_context_destroy_releases_device_memory():
    """csp_symbolic_destroy frees every device array of a context (workspaces, constraint tables, probe slots, the QR
    workspace): creating and dropping contexts does not shrink the free device memory."""
    import gc

    def cycle():
        symb, S, msk, L, Yh, cptr, cidx, cval = _kkt_qr_case("nested_mid", 12, 51, density=0.05)
        sys = KKTSystem(symb, cptr, cidx, cval, max_rhs=8, tnzcols=0.0)
        Ld, Yd = dev(symb, L), dev(symb, Yh)
        sys.factor(Ld, Yd)
        sys.factor_qr(Ld, Yd)
        Ad = dev(symb, S.project(np.eye(symb.n)))
        chordal.probe_cone(Ad, Ad, np.linspace(0.0, 0.1, 8), "d")        # reserves the probe slots and streams
        del sys, Ld, Yd, symb
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()

    cycle()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(4):
        cycle()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < (32 << 20), (free0, free1)


@pytest.mark.timeout(120)
def test_kkt_qr_rank_deficient_stack_fails_instead_of_looping():
    """More constraints than entries of V (20 diagonal constraints on 15 unknowns; the drivers reject such a problem
    up front, solvers.py:351-352): the stack has rank 15, every pass breaks down again.  kkt_qr_factor must give up
    with the failure code of a Cholesky breakdown (ArithmeticError in the shim) after a bounded number of shifted
    passes -- it used to plan three more passes at every breakdown, for ever (found by scratch/fuzz_ipm.py)."""
    symb, S, A, msk = setup("diag", 61)
    L = A.copy()
    orc.cholesky(S, L)
    Yh = L.copy()
    orc.projected_inverse(S, Yh)
    rng = np.random.default_rng(62)
    m = 20
    pos = np.flatnonzero(msk)
    cptr = np.arange(m + 1) * len(pos)
    cidx = np.tile(pos, m)
    cval = rng.standard_normal(m * len(pos))
    sys = KKTSystem(symb, cptr, cidx, cval, max_rhs=4, tnzcols=0.0)
    with pytest.raises(ArithmeticError):
        sys.factor_qr(dev(symb, L), dev(symb, Yh))
    # the context stays usable
    cptr2, cidx2, cval2 = problems.random_constraints(symb, 5, density=0.5, seed=63)
    sys2 = KKTSystem(symb, cptr2, cidx2, cval2, max_rhs=4, tnzcols=0.0)
    sys2.factor_qr(dev(symb, L), dev(symb, Yh))


@pytest.mark.parametrize("name", sorted(GPU_PATTERNS))
def test_dual_scaling_point_in_one_call(name):
    """csp_cholesky_projected_inverse (solvers.py:881-891 as ONE entry point, its independent stages on side streams) against
    the oracle's cholesky + projected_inverse, against the two separate library calls, and -- the caches it leaves behind:
    inverse-form factor, Y_AA blocks, their Cholesky factors -- through a whole KKT factor + solve right after it."""
    symb, S, A, msk = setup(name, 11)
    Lref = A.copy()
    orc.cholesky(S, Lref)
    Yref = Lref.copy()
    orc.projected_inverse(S, Yref)
    L = dev(symb, A)
    Y = cspmatrix(symb, torch.full((symb.blklen,), float("nan"), dtype=torch.float64, device="cuda"))   # output only
    chordal.cholesky_projected_inverse(L, Y)
    assert rel(host(L)[msk], Lref[msk]) < TOL and rel(host(Y)[msk], Yref[msk]) < TOL
    assert not np.isnan(host(Y)).any()
    L2 = dev(symb, A)
    chordal.cholesky(L2)
    Y2 = L2.copy()
    chordal.projected_inverse(Y2)
    assert rel(host(L), host(L2)) < 1e-13 and rel(host(Y), host(Y2)) < 1e-12
    m = 7
    cptr, cidx, cval = problems.random_constraints(symb, m, density=0.05, seed=9)
    K = orc.KKT(S, cptr, cidx, cval)
    Href = K.schur_factor(Lref, Yref)
    sys = KKTSystem(symb, cptr, cidx, cval, max_rhs=4)
    rng = np.random.default_rng(13)
    bx, by = rng.standard_normal(symb.blklen) * msk, rng.standard_normal(m)
    for lazy in (False, True):            # eager: potrf(H) inside factor(); deferred: beside the first Hessian of the first solve_
        L = dev(symb, A)
        Y = cspmatrix(symb, torch.empty(symb.blklen, dtype=torch.float64, device="cuda"))
        chordal.lazy_status(symb, lazy)
        try:
            chordal.cholesky_projected_inverse(L, Y)
            solve = sys.factor(L, Y)
            for kk in (1.0, 0.5):             # the second call finds H factored
                xr, yr = K.solve(Lref, Yref, Href, bx, by, kk)
                bxd, byd = dev(symb, bx), torch.from_numpy(by.copy()).cuda()
                solve(bxd, byd, kk)
                if lazy:
                    chordal.check_status(symb)
                assert rel(host(bxd)[msk], xr[msk]) < 1e-9 and rel(byd.cpu().numpy(), yr) < 1e-9
            assert rel(np.tril(sys.H.cpu().numpy().T), np.tril(Href)) < 1e-9
        finally:
            chordal.lazy_status(symb, False)
    # not positive definite: the error of cholesky, and the context stays usable
    bad = A.copy()
    bad[symb.blkptr[symb.Nsn // 2]] = -1.0
    with pytest.raises(ArithmeticError):
        chordal.cholesky_projected_inverse(dev(symb, bad), cspmatrix(symb, torch.empty(symb.blklen, dtype=torch.float64, device="cuda")))
    L = dev(symb, A)
    Y = cspmatrix(symb, torch.empty(symb.blklen, dtype=torch.float64, device="cuda"))
    chordal.cholesky_projected_inverse(L, Y, factors=False)
    assert rel(host(Y)[msk], Yref[msk]) < TOL


def test_deferred_potrf_of_the_schur_complement():
    """Under chordal.lazy_status kkt_schur_factor leaves H unfactored until its first reader: dense_potrs, csp_status and
    kkt_solve all find it; a Schur complement that is not positive definite is reported by the next check_status."""
    symb, S, A, msk = setup("nested_mid", 21)
    L = dev(symb, A)
    Y = cspmatrix(symb, torch.empty(symb.blklen, dtype=torch.float64, device="cuda"))
    chordal.cholesky_projected_inverse(L, Y)
    m = 5
    cptr, cidx, cval = problems.random_constraints(symb, m, density=0.05, seed=22)
    sys = KKTSystem(symb, cptr, cidx, cval, max_rhs=4)
    sys.factor(L, Y)
    Hfac = sys.H.clone()                                   # eager: factored
    rhs = torch.from_numpy(np.random.default_rng(23).standard_normal(m)).cuda()
    y0 = rhs.clone()
    sys._potrs(y0)
    chordal.lazy_status(symb, True)
    try:
        sys.factor(L, Y)                                   # H built, factorisation pending
        y1 = rhs.clone()
        sys._potrs(y1)                                     # the first reader factors it
        chordal.check_status(symb)
        assert float((y1 - y0).abs().max()) < 1e-12 * float(y0.abs().max())
        assert float((sys.H - Hfac).abs().max()) < 1e-12 * float(Hfac.abs().max())
        sys.factor(L, Y)
        chordal.check_status(symb)                         # ... and so does the status read-out
        assert float((sys.H - Hfac).abs().max()) < 1e-12 * float(Hfac.abs().max())
        # linearly dependent constraints: H singular, the deferred potrf must report it
        cptr2 = np.concatenate([cptr, [cptr[-1] + (cptr[1] - cptr[0])]])
        cidx2 = np.concatenate([cidx, cidx[cptr[0]:cptr[1]]])
        cval2 = np.concatenate([cval, cval[cptr[0]:cptr[1]]])
        sys2 = KKTSystem(symb, cptr2, cidx2, cval2, max_rhs=4)
        solve = sys2.factor(L, Y)
        bxd = dev(symb, np.random.default_rng(24).standard_normal(symb.blklen) * msk)
        byd = torch.from_numpy(np.random.default_rng(25).standard_normal(m + 1)).cuda()
        solve(bxd, byd, 1.0)
        with pytest.raises(ArithmeticError):
            chordal.check_status(symb)
        chordal.check_status(symb)                         # reported once
    finally:
        chordal.lazy_status(symb, False)


def test_two_systems_with_deferred_potrf_on_one_symbolic():
    """two KKT systems share a Symbolic (the drivers do that for kktsolver='qr'); under chordal.lazy_status each
    factor() leaves its H waiting for its factorisation, and the context has ONE slot for that.  Both systems factored, then
    solved in both orders, must give what they give alone; a system that dies while its H waits must not leave the context
    with a dangling pointer."""
    symb, S, A, msk = setup("nested_mid", 41)
    L = dev(symb, A)
    Y = cspmatrix(symb, torch.empty(symb.blklen, dtype=torch.float64, device="cuda"))
    chordal.cholesky_projected_inverse(L, Y)
    cons = [problems.random_constraints(symb, m, density=0.05, seed=sd) for m, sd in ((5, 42), (7, 43))]
    rng = np.random.default_rng(44)
    rhs = [(rng.standard_normal(symb.blklen) * msk, rng.standard_normal(len(c[0]) - 1)) for c in cons]
    # each system alone, eager
    alone = []
    for (cptr, cidx, cval), (bx, by) in zip(cons, rhs):
        sys = KKTSystem(symb, cptr, cidx, cval, max_rhs=4)
        solve = sys.factor(L, Y)
        bxd, byd = dev(symb, bx), torch.from_numpy(by.copy()).cuda()
        solve(bxd, byd, 1.0)
        alone.append((host(bxd), byd.cpu().numpy()))
    chordal.lazy_status(symb, True)
    try:
        for order in ((0, 1), (1, 0)):
            systems = [KKTSystem(symb, *c, max_rhs=4) for c in cons]
            solves = [systems[0].factor(L, Y), systems[1].factor(L, Y)]       # A pending, then B takes the slot
            for q in order:
                bxd, byd = dev(symb, rhs[q][0]), torch.from_numpy(rhs[q][1].copy()).cuda()
                solves[q](bxd, byd, 1.0)
                chordal.check_status(symb)
                assert rel(host(bxd)[msk], alone[q][0][msk]) < 1e-11, (order, q)
                assert rel(byd.cpu().numpy(), alone[q][1]) < 1e-11, (order, q)
        # a system garbage-collected with its H still waiting
        dead = KKTSystem(symb, *cons[0], max_rhs=4)
        dead.factor(L, Y)
        del dead
        import gc
        gc.collect()
        junk = torch.full((64,), float("nan"), dtype=torch.float64, device="cuda")      # may land in the freed H
        chordal.check_status(symb)                                # must not factor freed memory (no pending mark left)
        sys = KKTSystem(symb, *cons[1], max_rhs=4)
        solve = sys.factor(L, Y)
        bxd, byd = dev(symb, rhs[1][0]), torch.from_numpy(rhs[1][1].copy()).cuda()
        solve(bxd, byd, 1.0)
        chordal.check_status(symb)
        assert rel(byd.cpu().numpy(), alone[1][1]) < 1e-11
        assert bool(torch.isnan(junk).all())
    finally:
        chordal.lazy_status(symb, False)


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_kkt_step_under_delay_injection(seed):
    """Race hunt: the whole single-rank step -- scaling point in one call, Schur complement with the
    deferred potrf, two solve_ -- with a seeded random spin kernel behind every internal fork, before every join and before
    one launch in eight (csp_tune CSP_TUNE_RACE).  Every cross-stream edge that is missing shows as a changed result."""
    symb, S, A, msk = setup("nested_mid", 51)
    m = 40
    cptr, cidx, cval = problems.random_constraints(symb, m, density=0.002, seed=52)
    rng = np.random.default_rng(53)
    bx, by = rng.standard_normal(symb.blklen) * msk, rng.standard_normal(m)

    def step(lazy):
        L = dev(symb, A)
        Y = cspmatrix(symb, torch.empty(symb.blklen, dtype=torch.float64, device="cuda"))
        sys = KKTSystem(symb, cptr, cidx, cval, max_rhs=m, tnzcols=0.0)
        chordal.lazy_status(symb, lazy)
        try:
            chordal.cholesky_projected_inverse(L, Y)
            solve = sys.factor(L, Y)
            out = []
            for _ in range(2):
                bxd, byd = dev(symb, bx), torch.from_numpy(by.copy()).cuda()
                solve(bxd, byd, 1.0)
                out.append((host(bxd), byd.cpu().numpy()))
            if lazy:
                chordal.check_status(symb)
            return out, sys.H.cpu().numpy()
        finally:
            chordal.lazy_status(symb, False)

    ref, Href = step(False)
    chordal.tune(symb, chordal.TUNE_RACE, seed)
    n0 = chordal.race_injected(symb)
    try:
        for lazy in (True, False):
            got, H = step(lazy)
            assert rel(np.tril(H), np.tril(Href)) < 1e-11, (seed, lazy)
            for (x, y), (xr, yr) in zip(got, ref):
                assert rel(x[msk], xr[msk]) < 1e-10 and rel(y, yr) < 1e-10, (seed, lazy, rel(x[msk], xr[msk]), rel(y, yr))
    finally:
        chordal.tune(symb, chordal.TUNE_RACE, 0)
    assert chordal.race_injected(symb) - n0 >= 20          # the harness did run (forks, joins, one launch in eight)


def test_delay_injection_detects_a_removed_join():
    """The harness's own sensitivity: with the joins of the side branches removed (TUNE_RACE_DROP_JOINS) the deferred potrf(H)
    of kkt_solve is no longer ordered before potrs, and under delay injection the step must come out WRONG for at least one
    of a few seeds -- a harness that could not see a missing edge would prove nothing by passing."""
    symb, S, A, msk = setup("nested_mid", 61)
    m = 40
    cptr, cidx, cval = problems.random_constraints(symb, m, density=0.002, seed=62)
    rng = np.random.default_rng(63)
    bx, by = rng.standard_normal(symb.blklen) * msk, rng.standard_normal(m)
    sys = KKTSystem(symb, cptr, cidx, cval, max_rhs=m, tnzcols=0.0)

    def step():
        L = dev(symb, A)
        Y = cspmatrix(symb, torch.empty(symb.blklen, dtype=torch.float64, device="cuda"))
        chordal.cholesky_projected_inverse(L, Y)
        solve = sys.factor(L, Y)
        bxd, byd = dev(symb, bx), torch.from_numpy(by.copy()).cuda()
        solve(bxd, byd, 1.0)
        torch.cuda.synchronize()
        return byd.cpu().numpy()

    chordal.lazy_status(symb, True)
    try:
        ref = step()
        chordal.check_status(symb)
        changed = 0
        chordal.tune(symb, chordal.TUNE_RACE_DROP_JOINS, 1)
        try:
            for seed in range(71, 75):
                chordal.tune(symb, chordal.TUNE_RACE, seed | (2000 << 32))          # every side branch 2 ms late: potrs meets the raw H
                y = step()
                changed += int(not (rel(y, ref) < 1e-10))            # (NaN counts as changed)
        finally:
            chordal.tune(symb, chordal.TUNE_RACE, 0)
            chordal.tune(symb, chordal.TUNE_RACE_DROP_JOINS, 0)
        try:
            chordal.check_status(symb)                                # whatever the broken runs latched is discarded
        except ArithmeticError:
            pass
        assert changed >= 1
        assert rel(step(), ref) < 1e-12                               # and the context is sound again
        chordal.check_status(symb)
    finally:
        chordal.lazy_status(symb, False)


@pytest.mark.parametrize("name,m,density,fused", [("nested_mid", 40, 0.002, True), ("nested_mid", 40, 0.002, False),
                                                    ("fam_top", 36, 0.004, True)])
def test_family_updates_formed_by_the_extend_add(name, m, density, fused, monkeypatch):
    """Round 4: the family parents' update matrices are not written by the family sweep and read back by the extend-add of
    the front above -- that extend-add forms them itself from the families' tables and the static term lists, into the front
    it holds in LDS (k_lf_assemble_fz, front_famt.hip).  H, x, y against the oracle; the launch counters show which route ran;
    SMCP_FZ=0 (checked in a fresh context through csp_tune's sibling: the switch is read once per process, so the unfused
    reference here is the same problem with the closed-form leaf route off, which never requests the fusion)."""
    symb, S, A, msk = setup(name, 31)
    rng = np.random.default_rng(32)
    L = A.copy()
    orc.cholesky(S, L)
    Yh = L.copy()
    orc.projected_inverse(S, Yh)
    cptr, cidx, cval = problems.random_constraints(symb, m, density=density, seed=33)
    K = orc.KKT(S, cptr, cidx, cval)
    Href = K.schur_factor(L, Yh)
    sys = KKTSystem(symb, cptr, cidx, cval, max_rhs=m, tnzcols=0.0)
    chordal.tune(symb, chordal.TUNE_LEAFGRAM, 2 if fused else 0)
    try:
        Ld, Yd = dev(symb, L), dev(symb, Yh)
        box = {}
        counts = _launch_counts(symb, lambda: box.setdefault("solve", sys.factor(Ld, Yd)))
        if fused:
            assert counts.get("k_lf_assemble_fz", 0) >= 1 and counts.get("k_fam_terms", 0) >= 1, counts
        else:
            assert counts.get("k_lf_assemble_fz", 0) == 0, counts
        assert rel(np.tril(sys.H.cpu().numpy().T), np.tril(Href)) < 1e-9
        bx = rng.standard_normal(symb.blklen) * msk
        by = rng.standard_normal(m)
        xr, yr = K.solve(L, Yh, Href, bx, by, 0.5)
        bxd, byd = dev(symb, bx), torch.from_numpy(by.copy()).cuda()
        box["solve"](bxd, byd, 0.5)
        assert rel(host(bxd)[msk], xr[msk]) < 1e-9 and rel(byd.cpu().numpy(), yr) < 1e-9
    finally:
        chordal.tune(symb, chordal.TUNE_LEAFGRAM, 1)


def test_fused_extend_add_with_shared_pairs():
    """k_lf_assemble_fz with the children of EVERY (front, right-hand side) pair dealt over three workgroups -- each gathers its
    share into a front of its own and adds it to the cleared panel / update block with global atomics (k_lf_zero_pairs; one
    rank's owned sweep of an eight-rank job takes this route by itself, and the thin last round of the headline launch a
    variant of it).  SMCP_ALDS_Z=3 forces it; the switch is read once per process, so the fused cases above run again in a
    child interpreter."""
    import subprocess
    import sys
    env = dict(os.environ, SMCP_ALDS_Z="3")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-p", "no:cacheprovider",
                        "-k", "family_updates_formed_by_the_extend_add and True"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "2 passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def _band_constraint(symb, width, rng):
    """one constraint holding every entry of V within `width` of the diagonal (permuted coordinates): width 0 = a diagonal
    matrix -- the trace constraint of an SDP relaxation --, which alone gives a (15, 64) parent with eight (5, 31) leaves a
    (family, constraint) list of 15 + 8 x 5 = 55 entries; width 1: 101, width 4: 185"""
    cp, ri = symb.sparsity_pattern()
    col = np.repeat(np.arange(symb.n), np.diff(cp))
    pos = np.sort(symb.ccs_to_blk()[(ri - col) <= width])
    return pos.astype(np.int64), rng.standard_normal(len(pos))


@pytest.mark.parametrize("name,width,nlong,m,fz", [("nested_mid", 0, 1, 24, True), ("nested_mid", 1, 1, 24, True), ("nested_mid", 4, 1, 24, True),
                                                   ("nested_mid", 4, 2, 48, True), ("fam_odd", 4, 1, 24, False), ("fam_top", 2, 1, 24, True),
                                                   ("nested_mid", 4, -1, 3, None)])
def test_entry_driven_sweeps_with_long_lists_among_short_ones(name, width, nlong, m, fz):
    """Round 5: the entry-driven family sweeps take (family, constraint) lists of any length up to 384 in
    chunks -- 48 entries of descriptors per chunk in k_fam_terms, 64 terms (one per lane) in the fused extend-add -- and the route
    is chosen by the MEAN list length, not the longest: a trace constraint (55 entries per family) or a band (101, 185: two / four
    descriptor chunks, two / three term chunks) among sparse constraints no longer sends the whole set to the dense family sweep.
    H, x, y against the oracle; the launch counters say which route ran.  nlong = -1: every constraint is a band (mean beyond the
    gate): the dense route, same answers."""
    symb, S, A, msk = setup(name, 51)
    rng = np.random.default_rng(52)
    L = A.copy()
    orc.cholesky(S, L)
    Yh = L.copy()
    orc.projected_inverse(S, Yh)
    # (m: the fused extend-add wants 32 (front, right-hand side) pairs -- two top fronts x 24 --, and both entry-driven routes a mean
    # of at most four entries per (clique, constraint))
    cptr, cidx, cval = problems.random_constraints(symb, m, density=0.002, seed=53)
    cols = [(cidx[cptr[j]:cptr[j + 1]], cval[cptr[j]:cptr[j + 1]]) for j in range(m)]
    for j in (range(m) if nlong < 0 else [3, 17][:nlong]):
        cols[j] = _band_constraint(symb, width, rng)
    cptr = np.concatenate([[0], np.cumsum([len(p) for p, _ in cols])]).astype(np.int64)
    cidx = np.concatenate([p for p, _ in cols]).astype(np.int64)
    cval = np.concatenate([v for _, v in cols])
    K = orc.KKT(S, cptr, cidx, cval)
    Href = K.schur_factor(L, Yh)
    sys_ = KKTSystem(symb, cptr, cidx, cval, max_rhs=m, tnzcols=0.0)
    chordal.tune(symb, chordal.TUNE_LEAFGRAM, 2)
    try:
        Ld, Yd = dev(symb, L), dev(symb, Yh)
        box = {}
        counts = _launch_counts(symb, lambda: box.setdefault("solve", sys_.factor(Ld, Yd)))
        if fz is None:
            assert counts.get("k_fam_terms", 0) == 0 and counts.get("k_lf_assemble_fz", 0) == 0, counts
        else:
            assert counts.get("k_fam_terms", 0) >= 1, counts
            assert (counts.get("k_lf_assemble_fz", 0) >= 1) == fz, counts
        assert rel(np.tril(sys_.H.cpu().numpy().T), np.tril(Href)) < 1e-9
        bx = rng.standard_normal(symb.blklen) * msk
        by = rng.standard_normal(m)
        xr, yr = K.solve(L, Yh, Href, bx, by, 0.5)
        bxd, byd = dev(symb, bx), torch.from_numpy(by.copy()).cuda()
        box["solve"](bxd, byd, 0.5)
        assert rel(host(bxd)[msk], xr[msk]) < 1e-9 and rel(byd.cpu().numpy(), yr) < 1e-9
    finally:
        chordal.tune(symb, chordal.TUNE_LEAFGRAM, 1)

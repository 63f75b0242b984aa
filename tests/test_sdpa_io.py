"""Row N1: SDPA sparse reader / writer (reference: src/C/misc.c:56-365, src/python/base.py:71-86,177-217).

Hand-written fixtures under tests/golden/sdpa/ with the expected (A, b, blockstruct) spelled out here; no GPU.
The solves run the host drivers over the CPU oracle (tests/oracle_backend.py)."""
import math
import os

import numpy as np
import pytest
import scipy.sparse as sp

from smcp_amd import base, solvers
from tests.oracle_backend import oracle_backend

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sdpa")


@pytest.fixture(autouse=True)
def _quiet():
    saved = dict(solvers.options)
    solvers.options.update(show_progress=False, maxiters=100)
    yield
    solvers.options.clear()
    solvers.options.update(saved)


def dense_cols(A, n):
    """columns of A as dense symmetric n x n matrices (lower triangles stored)."""
    out = []
    for k in range(A.shape[1]):
        L = np.asarray(sp.csc_matrix(A[:, k]).todense()).reshape(n, n, order="F")
        assert np.allclose(np.triu(L, 1), 0.0), "only lower-triangular positions may be stored"
        out.append(L + np.tril(L, -1).T)
    return out


def test_sdplib_style_header_braces_diagonal_block_and_zero():
    """The reference skips any text that is not a digit or a sign between numbers (misc.c:94,176,185,205-209):
    '2 =mdim', '{2, -2}', trailing words.  Negative block size = diagonal block; explicit zeros dropped (misc.c:216)."""
    fn = os.path.join(GOLD, "sdplib_style.dat-s")
    assert base.sdpa_readhead(fn) == (4, 2, [2, -2])
    A, b, bs = base.sdpa_read(fn)
    assert bs == [2, -2] and A.shape == (16, 3)
    assert np.array_equal(b, [10.0, 20.0])
    assert A.nnz == 7                                      # 8 records, one explicit zero dropped
    F0, F1, F2 = dense_cols(A, 4)
    assert np.array_equal(F0, [[1.0, -0.5, 0, 0], [-0.5, 0, 0, 0], [0, 0, 3.0, 0], [0, 0, 0, 0]])
    assert np.array_equal(F1, [[0, 2.0, 0, 0], [2.0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 1.5]])
    assert np.array_equal(F2, [[0, 0, 0, 0], [0, -0.4, 0, 0], [0, 0, 2.5, 0], [0, 0, 0, 0]])
    An, bn, _ = base.sdpa_read(fn, neg=True)               # misc.c:186-187, 223-224
    assert abs(An + A).max() == 0 and np.array_equal(bn, -b)


def test_two_blocks_parentheses_and_lower_triangular_entries():
    fn = os.path.join(GOLD, "two_blocks.dat-s")
    A, b, bs = base.sdpa_read(fn)
    assert bs == [3, 2] and A.shape == (25, 4)
    assert np.array_equal(b, [1.0, -2.0, 0.5])
    F = dense_cols(A, 5)
    E = [np.zeros((5, 5)) for _ in range(4)]
    E[0][0, 0] = 2.0; E[0][0, 2] = E[0][2, 0] = -1.0; E[0][3, 4] = E[0][4, 3] = 0.25
    E[1][1, 1] = 1.0; E[1][3, 3] = 1.0
    E[2][2, 0] = E[2][0, 2] = 0.5; E[2][4, 4] = -1.0       # "2 1 3 1 0.5" is given below the diagonal
    E[3][1, 2] = E[3][2, 1] = 4.0
    for got, want in zip(F, E):
        assert np.array_equal(got, want)


def test_sdp_constructor_negates_like_the_reference(tmp_path):
    """SDP(filename) reads with neg=True (base.py:189,194) and write_sdpa writes with neg=True (base.py:212):
    an SDPA file states max <F0,Y> s.t. <Fi,Y> = ci, the solver minimises <C,X> with C = -F0, A_i = -F_i, b = -c."""
    fn = os.path.join(GOLD, "example1.dat-s")
    P = base.SDP(fn)
    assert (P.n, P.m, P.blockstruct) == (2, 3, [2])
    assert P._pname == "example1"
    assert np.array_equal(P.b, [-48.0, 8.0, -20.0])
    C = np.asarray(P.get_A(0).todense())
    assert np.array_equal(C, [[11.0, 0.0], [0.0, -23.0]])
    assert np.array_equal(np.asarray(P.get_A(3).todense()), [[0.0, 8.0], [8.0, 2.0]])
    # write -> read round trip through the reference's naming rule (fname + '.dat-s', refuses to overwrite)
    stem = str(tmp_path / "copy")
    P.write_sdpa(stem)
    with pytest.raises(IOError):
        P.write_sdpa(stem)
    A0, b0, _ = base.sdpa_read(fn)
    A1, b1, bs1 = base.sdpa_read(stem + ".dat-s")           # the file on disk holds the ORIGINAL (un-negated) data
    assert abs(A1 - A0).max() == 0 and np.array_equal(b1, b0) and bs1 == [2]
    Q = base.SDP(stem + ".dat-s")
    assert abs(sp.csc_matrix(Q.A) - sp.csc_matrix(P.A)).max() == 0 and np.array_equal(Q.b, P.b)
    # compressed variant and the pickle pair (base.py:219-270)
    P.write_sdpa(str(tmp_path / "z"), compress=True)
    Z = base.SDP(str(tmp_path / "z.dat-s.bz2"))
    assert Z._pname == "z" and abs(sp.csc_matrix(Z.A) - sp.csc_matrix(P.A)).max() == 0
    P.save(str(tmp_path / "p"))
    R = base.SDP(str(tmp_path / "p.pkl"))
    assert abs(sp.csc_matrix(R.A) - sp.csc_matrix(P.A)).max() == 0 and np.array_equal(R.b, P.b)
    with pytest.raises(NameError):
        base.SDP(str(tmp_path / "p.txt"))


def _kkt_certificate(P, sol, tol=1e-6):
    """Dense optimality conditions of min <C,X> s.t. <A_i,X> = b_i, X >= 0 and its dual."""
    n, m = P.n, P.m
    X = np.asarray(sol["x"].todense())
    S = np.asarray(sol["s"].todense())
    y = np.asarray(sol["y"]).reshape(-1)
    C = np.asarray(P.get_A(0).todense())
    Ai = [np.asarray(P.get_A(i + 1).todense()) for i in range(m)]
    assert np.allclose([np.sum(a * X) for a in Ai], P.b, atol=tol)
    assert np.linalg.norm(sum(yi * a for yi, a in zip(y, Ai)) + S - C) < tol
    assert np.linalg.eigvalsh(X).min() > -tol and np.linalg.eigvalsh(S).min() > -tol
    assert abs(np.sum(C * X) - P.b @ y) < 2e-5 * (1 + abs(P.b @ y))     # default reltol 1e-6 of the drivers
    return float(np.sum(C * X))


def test_known_optimum_sign_of_objective_maxeig():
    """2 x 2 largest-eigenvalue problem written in SDPA form: the SDPA dual optimum is +lambda_max, so the SDP the
    reference builds from the file (negated data) has optimum -lambda_max.  A reader without the negation would
    solve min <M,Y>, tr Y = 1 and return +lambda_min = 1.382 instead."""
    P = base.SDP(os.path.join(GOLD, "maxeig2.dat-s"))
    assert np.array_equal(np.asarray(P.get_A(0).todense()), [[-2.0, -1.0], [-1.0, -3.0]])
    assert np.array_equal(P.b, [-1.0])
    with oracle_backend():
        sol = P.solve_feas()
    assert sol["status"] == "optimal"
    lam = (5.0 + math.sqrt(5.0)) / 2.0
    assert abs(_kkt_certificate(P, sol) + lam) < 2e-5
    assert abs(sol["primal objective"] + lam) < 2e-5


def test_known_optimum_sdpa_manual_example1():
    """Example 1 of the SDPA user's manual (optimum 41.9 for both SDPA problems, which the manual prints as
    objValPrimal = objValDual = -4.19e+01 in its max-form sign convention): through SDP(filename) the negated
    problem min <-F0,X> s.t. <-Fi,X> = -ci has optimum +41.9; certified by its dense optimality conditions."""
    P = base.SDP(os.path.join(GOLD, "example1.dat-s"))
    with oracle_backend():
        sol = P.solve_feas()
        sol2 = P.solve_esd()
    assert sol["status"] == "optimal" and sol2["status"] == "optimal"
    obj = _kkt_certificate(P, sol)
    assert abs(obj - 41.9) < 2e-4
    assert abs(sol2["primal objective"] - 41.9) < 2e-4


def test_writer_header_matches_reference_layout(tmp_path):
    """misc.c:303-323: a comment line, 'm = m', 'nBlocks = nBlocks', the block sizes, b; entries as
    '<matno> <blkno> <i> <j> <value>' in the upper triangle, zeros skipped (misc.c:350)."""
    A = sp.csc_matrix((np.array([1.0, 0.0, 2.0, -3.0]), (np.array([0, 4, 1, 8]), np.array([0, 0, 1, 1]))), shape=(9, 2))
    fn = str(tmp_path / "w.dat-s")
    base.sdpa_write(fn, A, np.array([0.25]), [2, -1])
    lines = open(fn).read().splitlines()
    assert lines[0].startswith("*")
    assert lines[1] == "1 = m" and lines[2] == "2 = nBlocks" and lines[3] == "2 -1"
    assert [float(t) for t in lines[4].split()] == [0.25]
    assert lines[5:] == ["0 1 1 1 1", "1 1 1 2 2", "1 2 1 1 -3"]
    with pytest.raises(ValueError):                        # an entry that couples two blocks cannot be written
        base.sdpa_write(fn, sp.csc_matrix(([1.0], ([2], [0])), shape=(9, 2)), np.array([0.0]), [2, -1])


def test_band_roundtrip_is_exact(tmp_path):
    P = base.band_SDP(12, 4, 2, seed=3)
    P.write_sdpa(str(tmp_path / "p"))
    Q = base.SDP(str(tmp_path / "p.dat-s"))
    assert Q.n == P.n and Q.m == P.m
    assert abs(sp.csc_matrix(Q.A) - sp.csc_matrix(P.A)).max() == 0
    assert np.array_equal(Q.b, P.b)
    assert base.sdpa_readhead(str(tmp_path / "p.dat-s")) == (12, 4, [12])

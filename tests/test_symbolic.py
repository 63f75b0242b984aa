"""Host symbolic layer (C++ behind the C-ABI) against the pure-Python reference symbolic and
against structural known-answers from the reference's docs."""
import numpy as np
import pytest

from oracle.symbolic_ref import symbolic_ref
from smcp_amd import problems
from smcp_amd.symbolic import Symbolic, maxcardsearch, mindegree
from tests.helpers import PATTERNS, edges_of


def clique_sets(p, rowptr, rowidx):
    return {frozenset(p[rowidx[rowptr[k]:rowptr[k + 1]]].tolist()) for k in range(len(rowptr) - 1)}


@pytest.mark.parametrize("name", sorted(PATTERNS))
def test_matches_reference_symbolic(name):
    pat = PATTERNS[name]()
    s = Symbolic(pat)
    r = symbolic_ref(pat[0], edges_of(pat))
    assert s.fill == 0 and r["fill"] == 0          # chordal input in a perfect elimination order
    assert s.nnz == r["nnz"]
    assert s.Nsn == len(r["snptr"]) - 1
    # same set of maximal cliques in ORIGINAL labels (tie-breaking of orderings may differ)
    assert clique_sets(s.p, s.rowptr, s.rowidx) == clique_sets(r["p"], r["rowptr"], r["rowidx"])


@pytest.mark.parametrize("name", sorted(PATTERNS))
def test_structure_invariants(name):
    s = Symbolic(PATTERNS[name]())
    nn, na = s.clique_sizes()
    assert nn.sum() == s.n
    assert (np.sort(s.p) == np.arange(s.n)).all()
    assert s.blklen == int(((nn + na) * nn).sum()) and s.updlen == int((na * na).sum())
    cl = s.cliques()
    for k in range(s.Nsn):
        rows = cl[k]
        assert (np.diff(rows) > 0).all()
        assert (rows[:nn[k]] == np.arange(s.snptr[k], s.snptr[k + 1])).all()
        pk = s.snpar[k]
        if pk >= 0:
            assert pk > k                                   # postorder: children before parents
            rel = s.relidx[s.sepptr[k]:s.sepptr[k + 1]]
            assert (cl[pk][rel] == rows[nn[k]:]).all()       # separator inside parent's clique
        else:
            assert na[k] == 0
    # level sets: every parent strictly above its children, every clique listed once
    lev = np.empty(s.Nsn, dtype=int)
    for l in range(s.nlev):
        lev[s.levidx[s.levptr[l]:s.levptr[l + 1]]] = l
    for k in range(s.Nsn):
        if s.snpar[k] >= 0:
            assert lev[s.snpar[k]] > lev[k]
    assert sorted(s.levidx.tolist()) == list(range(s.Nsn))


def test_config_shapes():
    """Clique counts of the BASELINE configurations."""
    s = Symbolic(problems.band_pattern(200, 3))
    assert (s.Nsn, s.nnz, s.blklen) == (197, 794, 800)       # cfg1: (1,3)x196 + (4,0)
    s = Symbolic(problems.block_arrow_pattern(20, 64, 128))
    nn, na = s.clique_sizes()
    # maximal supernodes (Pothen-Sun, as CHOMPACK): one block merges with the arrow head, so the
    # cliques are exactly the maximal cliques block_i U head: 19 x (64,128) + 1 x (192,0)
    from collections import Counter
    assert Counter(zip(nn.tolist(), na.tolist())) == {(64, 128): 19, (192, 0): 1}
    # reference docs: band_SDP(n=100, bw=2) has nnz = 297 (docs index.rst:605-612)
    assert Symbolic(problems.band_pattern(100, 2)).nnz == 297


def test_synth50k_shape():
    s = Symbolic(problems.nested_block_arrow_pattern())
    assert s.n == 50000 and s.Nsn == 8073 and s.fill == 0
    nn, na = s.clique_sizes()
    from collections import Counter
    c = Counter(zip(nn.tolist(), na.tolist()))
    assert c == {(5, 31): 7168, (15, 64): 896, (64, 128): 8, (208, 0): 1}
    assert s.nlev == 4


def test_nonchordal_fill_and_orderings():
    # 5x5 grid graph is not chordal: symbolic must report fill; min-degree beats natural order
    g = 6
    idx = lambda i, j: i * g + j
    E = [(idx(i, j), idx(i, j + 1)) for i in range(g) for j in range(g - 1)]
    E += [(idx(i, j), idx(i + 1, j)) for i in range(g - 1) for j in range(g)]
    n = g * g
    I = np.array([max(a, b) for a, b in E] + list(range(n)))
    J = np.array([min(a, b) for a, b in E] + list(range(n)))
    order = np.lexsort((I, J))
    cp = np.zeros(n + 1, dtype=np.int64)
    np.add.at(cp, J + 1, 1)
    pat = (n, np.cumsum(cp), I[order].astype(np.int64))
    s0 = Symbolic(pat)
    assert s0.fill > 0
    r = symbolic_ref(n, E)
    assert s0.nnz == r["nnz"]
    p = mindegree(pat)
    s1 = Symbolic(pat, p)
    assert s1.fill <= s0.fill
    r1 = symbolic_ref(n, E, p)
    assert s1.nnz == r1["nnz"]


def test_maxcardsearch_is_peo_for_chordal():
    rng = np.random.default_rng(0)
    n, cp, ri = problems.random_chordal_pattern(15, seed=5)
    # scramble labels so the given order is NOT a perfect elimination ordering
    q = rng.permutation(n)
    cols = np.repeat(np.arange(n), np.diff(cp))
    I, J = q[ri], q[cols]
    lo, hi = np.minimum(I, J), np.maximum(I, J)
    order = np.lexsort((hi, lo))
    c2 = np.zeros(n + 1, dtype=np.int64)
    np.add.at(c2, lo + 1, 1)
    pat = (n, np.cumsum(c2), hi[order].astype(np.int64))
    p = maxcardsearch(pat)
    assert Symbolic(pat, p).fill == 0


def test_index_map_roundtrip():
    pat = PATTERNS["rand2"]()
    s = Symbolic(pat)
    n, cp, ri = pat
    cols = np.repeat(np.arange(n), np.diff(cp))
    pos = s.index_map(ri, cols)
    assert (pos >= 0).all() and len(np.unique(pos)) == len(pos)
    assert (np.sort(pos) == np.sort(s.ccs_to_blk())).all()
    assert (s.index_map(cols, ri) == pos).all()          # symmetric lookup


@pytest.mark.parametrize("name", sorted(PATTERNS))
def test_replicated_symbolic_is_k_independent_copies(name):
    """csp_symbolic_replicate (the trial forest of the step-length searches): copy t of every index array is the
    base array shifted by t times the base totals; levels are the union of the copies' levels."""
    s = Symbolic(PATTERNS[name]())
    K = 3
    F = s.replicate(K)
    assert (F.n, F.nnz, F.Nsn, F.blklen, F.updlen, F.nlev) == (K * s.n, K * s.nnz, K * s.Nsn, K * s.blklen, K * s.updlen, s.nlev)
    assert (F.max_nn, F.max_na, F.max_front) == (s.max_nn, s.max_na, s.max_front)
    for name_, tot in (("snptr", s.n), ("blkptr", s.blklen), ("updptr", s.updlen), ("rowptr", len(s.rowidx)),
                       ("sepptr", len(s.relidx)), ("chptr", len(s.chidx)), ("ccsptr", s.nnz)):
        a, f = getattr(s, name_), getattr(F, name_)
        assert np.array_equal(f, np.concatenate([a[:-1] + t * tot for t in range(K)] + [[K * tot]])), name_
    assert np.array_equal(F.rowidx, np.concatenate([s.rowidx + t * s.n for t in range(K)]))
    assert np.array_equal(F.relidx, np.tile(s.relidx, K))
    assert np.array_equal(F.chidx, np.concatenate([s.chidx + t * s.Nsn for t in range(K)]))
    assert np.array_equal(F.snpar, np.concatenate([np.where(s.snpar >= 0, s.snpar + t * s.Nsn, -1) for t in range(K)]))
    assert np.array_equal(F.p, np.concatenate([np.asarray(s.p) + t * s.n for t in range(K)]))
    for l in range(s.nlev):
        base = s.levidx[s.levptr[l]:s.levptr[l + 1]]
        got = F.levidx[F.levptr[l]:F.levptr[l + 1]]
        assert sorted(got.tolist()) == sorted(np.concatenate([base + t * s.Nsn for t in range(K)]).tolist())
    with pytest.raises(ValueError):
        F.replicate(2)                                  # a forest is not replicated again

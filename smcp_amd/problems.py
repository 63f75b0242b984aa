"""Seeded synthetic sparsity patterns / problem data for the BASELINE.json configurations.

The reference's generators (base.py:514-949: mk_rand, band_SDP, mtxnorm_SDP, rand_SDP) draw from
cvxopt's RNG, which is unavailable here, so these are this repo's own numpy ``default_rng``
generators producing the same problem *shapes* (SURVEY.md 8d table).
All patterns are returned as (n, colptr, rowind) int64 lower-triangular CCS including the diagonal.
"""
import numpy as np


def _from_cliques(n, cliques):
    """Lower CCS pattern of the union of dense cliques; cliques = list of (cols, rows) index arrays,
    meaning all entries (r, c) with r in rows, c in cols, r >= c."""
    I, J = [], []
    for cols, rows in cliques:
        cc, rr = np.meshgrid(cols, rows, indexing="ij")
        m = rr >= cc
        I.append(rr[m])
        J.append(cc[m])
    I = np.concatenate(I).astype(np.int64)
    J = np.concatenate(J).astype(np.int64)
    key = np.unique(J * n + I)
    J, I = key // n, key % n
    cp = np.zeros(n + 1, dtype=np.int64)
    np.add.at(cp, J + 1, 1)
    return n, np.cumsum(cp), I


def band_pattern(n, w):
    """Banded pattern with half-bandwidth w (config 1: n=200, w=3; cf. band_SDP, base.py:563-636)."""
    cl = [(np.array([j]), np.arange(j, min(n, j + w + 1))) for j in range(n)]
    return _from_cliques(n, cl)


def block_arrow_pattern(nblocks, bs, arrow):
    """nblocks dense bs x bs diagonal blocks, each fully coupled to a dense arrow x arrow head
    (config 3: 2000 x 64 + 128).  Vectorised: no Python loop over blocks."""
    n = nblocks * bs + arrow
    # within-block lower triangles
    bi, bj = np.tril_indices(bs)
    off = (np.arange(nblocks) * bs)[:, None]
    I = [(off + bi[None, :]).ravel()]
    J = [(off + bj[None, :]).ravel()]
    # arrow rows x all block columns
    ar = np.arange(nblocks * bs, n)
    cols = np.arange(nblocks * bs)
    I.append(np.tile(ar, len(cols)))
    J.append(np.repeat(cols, arrow))
    ai, aj = np.tril_indices(arrow)
    I.append(nblocks * bs + ai)
    J.append(nblocks * bs + aj)
    I = np.concatenate(I).astype(np.int64)
    J = np.concatenate(J).astype(np.int64)
    order = np.lexsort((I, J))
    I, J = I[order], J[order]
    cp = np.zeros(n + 1, dtype=np.int64)
    np.add.at(cp, J + 1, 1)
    return n, np.cumsum(cp), I


def nested_block_arrow_pattern(nsub=8, nmid=112, nleaf_per_mid=8, leaf=(5, 31), mid=(15, 64), top=(64, 128),
                               root=208, seed=0, shared_mid_sep=False):
    """Three-level nested block-arrow pattern ("synth50k", config 5 with the defaults: n = 50000,
    8073 cliques): nsub subtrees x [1 x top + nmid x mid + nmid*nleaf_per_mid x leaf] + one root.
    A clique (nn, na) owns nn new columns and is coupled to na rows of its parent's clique -- a random subset of them, or,
    shared_mid_sep, for the mid cliques the FIRST na rows of the top clique (siblings with one separator: a block-arrow
    inside every subtree)."""
    rng = np.random.default_rng(seed)
    ln, la = leaf
    mn, ma = mid
    tn, ta = top
    n = nsub * (tn + nmid * (mn + nleaf_per_mid * ln)) + root
    cliques = []
    pos = 0
    root_cols = np.arange(n - root, n)
    assert ta <= root and ma <= tn + ta and la <= mn + ma
    for s in range(nsub):
        # layout inside a subtree: leaves of mid 0, mid 0, leaves of mid 1, mid 1, ..., top
        sub_start = pos
        top_own = np.arange(0)  # placeholder, fixed after counting
        sub_len = tn + nmid * (mn + nleaf_per_mid * ln)
        top_own = np.arange(sub_start + sub_len - tn, sub_start + sub_len)
        top_sep = np.sort(rng.choice(root_cols, size=ta, replace=False))
        top_clique = np.concatenate([top_own, top_sep])
        for mi in range(nmid):
            mid_own = np.arange(pos + nleaf_per_mid * ln, pos + nleaf_per_mid * ln + mn)
            mid_sep = np.sort(rng.choice(top_clique, size=ma, replace=False))
            if shared_mid_sep:
                mid_sep = np.sort(top_clique[:ma])
            mid_clique = np.concatenate([mid_own, mid_sep])
            for li in range(nleaf_per_mid):
                leaf_own = np.arange(pos, pos + ln)
                leaf_sep = np.sort(rng.choice(mid_clique, size=la, replace=False))
                cliques.append((leaf_own, np.concatenate([leaf_own, leaf_sep])))
                pos += ln
            cliques.append((mid_own, mid_clique))
            pos += mn
        cliques.append((top_own, top_clique))
        pos += tn
    cliques.append((root_cols, root_cols))
    return _from_cliques(n, cliques)


def random_chordal_pattern(nclq, max_nn=6, max_na=8, seed=0):
    """Random clique tree: each new clique hangs off a random earlier clique, keeps a random
    subset of it as separator and adds 1..max_nn new vertices. Children are numbered before parents."""
    rng = np.random.default_rng(seed)
    cl = []      # list of vertex lists (creation labels, root first)
    own = []
    nv = 0
    for t in range(nclq):
        nn = int(rng.integers(1, max_nn + 1))
        new = list(range(nv, nv + nn))
        nv += nn
        if t == 0:
            sep = []
        else:
            par = cl[int(rng.integers(0, t))]
            na = int(rng.integers(1, min(max_na, len(par)) + 1))
            sep = list(rng.choice(par, size=na, replace=False))
        cl.append(new + sep)
        own.append(new)
    n = nv
    relabel = lambda v: n - 1 - np.asarray(v, dtype=np.int64)  # later-created vertices eliminated first
    cliques = [(np.sort(relabel(o)), np.sort(relabel(c))) for o, c in zip(own, cl)]
    # all entries among clique members (not only own columns)
    full = [(c, c) for _, c in cliques]
    return _from_cliques(n, full)


def maxcut_graph_pattern(n=1000, nedges=5909, seed=0):
    """Random graph with the size of SDPLIB maxG51 (config 4); returns pattern and the edge list."""
    rng = np.random.default_rng(seed)
    edges = set()
    while len(edges) < nedges:
        i, j = rng.integers(0, n, size=2)
        if i != j:
            edges.add((max(i, j), min(i, j)))
    e = np.array(sorted(edges), dtype=np.int64)
    I = np.concatenate([e[:, 0], np.arange(n)])
    J = np.concatenate([e[:, 1], np.arange(n)])
    order = np.lexsort((I, J))
    I, J = I[order], J[order]
    cp = np.zeros(n + 1, dtype=np.int64)
    np.add.at(cp, J + 1, 1)
    return (n, np.cumsum(cp), I), e


def random_factor_blkval(symb, seed=0, offdiag=0.3):
    """Random lower-triangular factor with pattern V in blkval layout (host numpy): unit-ish
    positive diagonal, small off-diagonals, so that S = L L^T is comfortably positive definite."""
    rng = np.random.default_rng(seed)
    blk = rng.standard_normal(symb.blklen) * offdiag
    nn = np.diff(symb.snptr)
    nf = np.diff(symb.rowptr)
    bp = symb.blkptr
    for k in range(symb.Nsn):
        b = blk[bp[k]:bp[k + 1]].reshape((nf[k], nn[k]), order="F")
        b /= np.sqrt(nf[k])
        iu = np.triu_indices(nn[k], 1)
        b[:nn[k], :nn[k]][iu] = 0.0
        b[np.arange(nn[k]), np.arange(nn[k])] = 1.0 + rng.random(nn[k])
    return blk


def random_constraints(symb, m, density=0.005, seed=0, dense_on_v=False):
    """m sparse symmetric constraint matrices on V in blkval coordinates (CSC by constraint).
    nnz per constraint = max(1, round(density*|V|)) (the reference's UFSMC recipe, doc bench index.rst:479),
    or dense on V (as band_SDP, base.py:617-632) when dense_on_v."""
    rng = np.random.default_rng(seed)
    valid = lower_positions(symb)
    if dense_on_v:
        per = len(valid)
    else:
        per = max(1, int(round(density * symb.nnz)))
    cptr = np.arange(m + 1, dtype=np.int64) * per
    if dense_on_v:
        cidx = np.tile(valid, m)
    else:
        cidx = np.concatenate([np.sort(rng.choice(valid, size=per, replace=False)) for _ in range(m)])
    cval = rng.standard_normal(m * per)
    return cptr, cidx.astype(np.int64), cval


def lower_positions(symb):
    """blkval positions that belong to V (excludes the strict upper triangles of the NN blocks)."""
    return np.sort(symb.ccs_to_blk())

"""Shared test utilities: patterns, dense references (numpy), oracle wiring."""
import numpy as np

from oracle import oracle as orc
from smcp_amd import problems
from smcp_amd.symbolic import Symbolic


def edges_of(pat):
    n, cp, ri = pat
    cols = np.repeat(np.arange(n), np.diff(cp))
    return list(zip(ri.tolist(), cols.tolist()))


PATTERNS = {
    "band": lambda: problems.band_pattern(30, 3),
    "arrow": lambda: problems.block_arrow_pattern(6, 4, 5),
    "nested": lambda: problems.nested_block_arrow_pattern(nsub=2, nmid=3, nleaf_per_mid=2, leaf=(2, 4),
                                                          mid=(3, 5), top=(4, 6), root=8, seed=1),
    "rand1": lambda: problems.random_chordal_pattern(12, seed=1),
    "rand2": lambda: problems.random_chordal_pattern(25, max_nn=4, max_na=6, seed=2),
    "dense": lambda: problems.block_arrow_pattern(1, 7, 0) if False else problems.band_pattern(9, 8),
}


def make(name):
    pat = PATTERNS[name]()
    symb = Symbolic(pat)
    S = orc.Sym(symb)
    return pat, symb, S


def random_spd_on_V(S, seed=0):
    """Dense SPD matrix (permuted coordinates) whose sparsity pattern is exactly V:
    built as L L^T with L lower with pattern V (zero fill because V is chordal + PEO)."""
    rng = np.random.default_rng(seed)
    mask = np.tril(S.mask())
    L = np.where(mask, rng.standard_normal((S.n, S.n)) * 0.4, 0.0)
    L[np.diag_indices(S.n)] = 1.0 + rng.random(S.n)
    return L @ L.T, L


def proj(S, M):
    return np.where(S.mask(), M, 0.0)

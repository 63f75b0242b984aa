"""Subtree-sharded Schur complement on the HIP path with two ranks sharing ONE GPU (gloo backend; the
collectives are staged through the host): checks the device-side pieces the CPU gloo test cannot --
csp_set_partition, kkt_gram_sweep by clique set, csp_exchange_copy, kkt_gram_accumulate by range."""
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


@pytest.mark.parametrize("mode", ["columns", "subtree"])
def test_two_ranks_one_gpu(mode):
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
    err, spread = out.get()
    assert err < 1e-11 and spread < 1e-12

"""Whole subtree-sharded step of ONE rank of an N-rank job, on one GPU, through the production host code
(ShardedSchur.factor_scaling / factor / solve_) -- what tools/shard_budget.py cannot show: the host logic between
the phases, the status read-backs and the collective launches.

The collectives run over RCCL with a group of ONE rank (launch cost, no wire time); the all-gathers fill the other
ranks' slots with zeros, i.e. the other subtrees contribute no update to the top.  The numbers are then NOT a
solution of the problem (the top sees only this rank's subtree), but every kernel runs with the grid sizes and data
volumes of rank `--rank` of an N-rank job, and nothing fails: a missing update makes the top's front more positive
definite, not less.

--mode columns emulates the OTHER partitioning the same way: factorisation replicated (cholesky +
projected_inverse of the whole tree on every rank), the Schur complement sharded BY CONSTRAINT -- this rank's m / N columns
through the batched two-sweep csp_hessian + Amap (kkt_schur_columns) -- ONE all-reduce of H (80 KB), potrf and solve_
replicated.  No 53 MB gather, one collective per step; every kernel runs on the whole tree.

  python tools/shard_step_emul.py [--mode subtree|columns|both] [--worlds 1,2,4,8] [--steps 10]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_workload                      # noqa: E402
from smcp_amd import chordal, problems               # noqa: E402
from smcp_amd import kkt as kktmod                    # noqa: E402
from smcp_amd.cspmatrix import cspmatrix              # noqa: E402
from smcp_amd.kkt import KKTSystem                    # noqa: E402
from smcp_amd.symbolic import Symbolic                # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="synth50k")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--mode", default="both", choices=("subtree", "columns", "both"))
    ap.add_argument("--defer", type=int, default=1, help="1: the regime bench.py runs with N > 1 (status with H's all-reduce, deferred failure reports, x left sharded)")
    args = ap.parse_args()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    pat, m, density, label = build_workload(args.workload)
    symb = Symbolic(pat)
    cptr, cidx, cval = problems.random_constraints(symb, m, density=density, seed=1)
    S = cspmatrix(symb, torch.from_numpy(problems.random_factor_blkval(symb, seed=0)).cuda())
    kkt = KKTSystem(symb, cptr, cidx, cval, max_rhs=m)
    chordal.llt(S)
    msk = np.zeros(symb.blklen, dtype=bool)
    msk[symb.ccs_to_blk()] = True
    bx0 = torch.from_numpy(np.random.default_rng(2).standard_normal(symb.blklen) * msk).cuda()
    by0 = torch.from_numpy(np.random.default_rng(3).standard_normal(m)).cuda()
    G = dist.group.WORLD
    state = dict(world=1, rank=0)

    def gather(recv, send, group):                      # own slot filled, the others zero; one real (1-rank) collective
        recv.zero_()
        n = send.numel()
        r = state["rank"]
        dist.all_gather_into_tensor(recv[r * n:(r + 1) * n], send, group=G)

    kktmod._all_gather_into = gather
    kkt._world = lambda group: (state["world"], state["rank"])
    kkt.force_sharded = True
    out = {}
    lib = __import__("smcp_amd._lib", fromlist=["lib"]).lib()
    if args.mode in ("columns", "both"):
        L, Y = S.copy(), S.copy()
        H = kkt.H
        stq = lambda: torch.cuda.current_stream().cuda_stream
        for world in [int(w) for w in args.worlds.split(",")]:
            state.update(world=world, rank=min(args.rank, world - 1))
            kkt.partition = None
            kkt.force_sharded = False
            bx = cspmatrix(symb, bx0.clone())
            by = by0.clone()

            def cstep():
                bx.blkval.copy_(bx0)
                by.copy_(by0)
                L.blkval.copy_(S.blkval)
                chordal.cholesky(L)
                Y.blkval.copy_(L.blkval)
                chordal.projected_inverse(Y)
                kkt.build_schur(L, Y, G)          # columns j0 .. j1 of this rank + the all-reduce of H (world 1: all of them, Gram route)
                if state["world"] == 1 and "H" not in state:
                    state["H"] = H.clone()
                elif state["world"] > 1:
                    H.copy_(state["H"])           # what the other ranks' columns would have added (an 80 KB copy stands in for the wire)
                kkt._potrf()
                rc = lib.kkt_solve(symb.handle, L.blkval.data_ptr(), Y.blkval.data_ptr(), H.data_ptr(), m, 1.0,
                                   bx.blkval.data_ptr(), by.data_ptr(), stq())
                assert rc == 0, rc
                chordal.check_status(symb)

            chordal.lazy_status(symb, True)
            for _ in range(2):
                cstep()
            torch.cuda.synchronize()
            c0 = kkt.collectives
            t0 = time.perf_counter()
            for _ in range(args.steps):
                cstep()
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / args.steps
            chordal.lazy_status(symb, False)
            out["columns_%d" % world] = dict(ms_per_step=round(ms, 3), collectives_per_step=(kkt.collectives - c0) / args.steps,
                                             rank=state["rank"], columns=[(m * state["rank"]) // world, (m * (state["rank"] + 1)) // world])
            print("columns: world %d rank %d: %.3f ms per step (%.1f collectives)" % (world, state["rank"], ms,
                  out["columns_%d" % world]["collectives_per_step"]), flush=True)
        kkt.force_sharded = True
    for world in ([int(w) for w in args.worlds.split(",")] if args.mode in ("subtree", "both") else []):
        state.update(world=world, rank=min(args.rank, world - 1))
        kkt._install_partition(world, state["rank"])
        bx = cspmatrix(symb, bx0.clone())
        by = by0.clone()

        def step():
            bx.blkval.copy_(bx0)
            by.copy_(by0)
            if args.defer:
                Ls, Ys = kkt.factor_scaling(S, G, defer_status=True)
            else:
                Ls, Ys = kkt.factor_scaling(S, G)
            kkt.factor(Ls, Ys, G)(bx, by, 1.0, complete=not args.defer)
            if args.defer:
                chordal.check_status(symb)

        if args.defer:
            chordal.lazy_status(symb, True)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        c0 = kkt.collectives
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / args.steps
        out["subtree_%d" % world] = dict(ms_per_step=round(ms, 3), collectives_per_step=(kkt.collectives - c0) / args.steps,
                          rank=state["rank"])
        if args.defer:
            chordal.lazy_status(symb, False)
        print("subtree: world %d rank %d: %.3f ms per step (%.1f collectives)" % (world, state["rank"], ms, out["subtree_%d" % world]["collectives_per_step"]),
              flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/shard_step_emul.json", "w"), indent=1)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

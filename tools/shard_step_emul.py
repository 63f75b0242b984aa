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
    ap.add_argument("--top", default="replicated", choices=("replicated", "constraint", "both"),
                    help="subtree mode: the top swept by every rank for all constraints (all-gather of the root blocks) or sharded by constraint "
                         "(all-to-all by share + gather of the top panels on rank 0)")
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

    def a2a(recv, send, group):                         # own slice exchanged with itself through a real one-rank collective, the rest zero
        recv.zero_()
        w = send.numel() // state["world"]
        r = state["rank"]
        dist.all_to_all_single(recv[r * w:(r + 1) * w], send[r * w:(r + 1) * w], group=G)

    def gather_to(dst, recv_list, send, group):         # rank 0 of N: its own panels only; the others' slots stay as they are
        if state["rank"] == dst:
            dist.all_gather_into_tensor(recv_list[dst], send, group=G)
        else:
            tmp = torch.empty_like(send)
            dist.all_gather_into_tensor(tmp, send, group=G)

    kktmod._all_to_all = a2a
    kktmod._gather_to = gather_to
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
    tops = ["replicated", "constraint"] if args.top == "both" else [args.top]
    for top, world in ([(t, int(w)) for t in tops for w in args.worlds.split(",")] if args.mode in ("subtree", "both") else []):
        state.update(world=world, rank=min(args.rank, world - 1))
        kkt._install_partition(world, state["rank"])
        kkt.top_by_constraint = top == "constraint"        # (after the partition: its default depends on the world size)
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
        if top == "constraint" and world > 1:
            # what rank 0 RECEIVES in the gather of the top panels must be data H stays positive definite with: one step with the
            # replicated top fills the top's rows of the swept stack for every constraint, and the receive slots of the other
            # ranks are pre-filled from them (the emulated gather leaves those slots alone; the unpack rewrites the same values)
            kkt.top_by_constraint = False
            step()
            kkt.top_by_constraint = True
            from smcp_amd.kkt import column_range
            P = kkt.partition
            toplen = sum(b - a for a, b in P.top_ranges)
            width = max(column_range(m, d, world)[1] - column_range(m, d, world)[0] for d in range(world)) * toplen
            bufs = kkt.__dict__.setdefault("_xchg", {})
            send = torch.zeros(width, dtype=torch.float64, device="cuda")
            recv = [torch.zeros(width, dtype=torch.float64, device="cuda") for _ in range(world)] if state["rank"] == 0 else None
            if recv is not None:
                for r in range(1, world):
                    r0, r1 = column_range(m, r, world)
                    off = 0
                    for a, b in P.top_ranges:
                        kkt._stack_rows(0, r0, r1, a, b, recv[r][off:off + (r1 - r0) * (b - a)])
                        off += (r1 - r0) * (b - a)
            bufs[("top", width)] = (send, recv)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        c0 = kkt.collectives
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / args.steps
        skey = "subtree_%d" % world if top == "replicated" else "subtree_topshare_%d" % world
        out[skey] = dict(ms_per_step=round(ms, 3), collectives_per_step=(kkt.collectives - c0) / args.steps, rank=state["rank"], top=top)
        if args.defer:
            chordal.lazy_status(symb, False)
        print("subtree (top %s): world %d rank %d: %.3f ms per step (%.1f collectives)" % (top, world, state["rank"], ms, out[skey]["collectives_per_step"]),
              flush=True)
    # ---- wire model + SCALE-shaped lines.  NOT a measurement: rank 0's kernels and host logic are timed above
    # with one-rank collectives (launch cost, no wire time); the wire is modelled from the bytes every rank has to RECEIVE.
    # xGMI is point-to-point (7 links x ~153 GB/s per GPU, MI355X_MICROARCH.md): in an all-gather the N - 1 peers' chunks arrive over
    # N - 1 different links at once, so the time is one chunk over one link at LINK_EFF of its peak, plus a per-collective latency;
    # an all-reduce of a small vector (H: 80 KB, Amap: 800 B) is latency only.
    LINK_GBS, LINK_EFF, COLL_LAT_US = 153.0, 0.7, 15.0
    lines = []
    if args.mode in ("subtree", "both"):
        for top, world in [(t, int(w)) for t in tops for w in args.worlds.split(",")]:
            key = "subtree_%d" % world if top == "replicated" else "subtree_topshare_%d" % world
            if key not in out:
                continue
            state.update(world=world, rank=min(args.rank, world - 1))
            kkt._install_partition(world, state["rank"])
            _, sizes1, width1 = kkt._exchange_plan(G, 1)
            _, sizesm, widthm = kkt._exchange_plan(G, m)
            chunk = lambda width: 8.0 * width / (LINK_EFF * LINK_GBS * 1e9) * 1e6 if world > 1 else 0.0          # us: one peer's chunk over one link
            if top == "replicated":
                colls = [("cholesky exchange (all-gather)", chunk(width1)), ("Schur sweeps exchange (all-gather, %d right-hand sides)" % m, chunk(widthm)),
                         ("H all-reduce", 0.0), ("first Hessian exchange (all-gather)", chunk(width1)), ("Amap all-reduce", 0.0)]
            else:
                share = -(-m // world)
                toplen = sum(b - a for a, b in kkt.partition.top_ranges)
                colls = [("cholesky exchange (all-gather)", chunk(width1)),
                         ("Schur sweeps exchange (all-to-all by constraint share, %d right-hand sides per rank)" % share, chunk(width1 * share)),
                         ("top panels of the shares -> rank 0 (gather)", chunk(share * toplen)),
                         ("H all-reduce", 0.0), ("first Hessian exchange (all-gather)", chunk(width1)),
                         ("second Hessian exchange (all-gather)", chunk(width1)), ("Amap all-reduce", 0.0)]
            wire_us = sum(t + (COLL_LAT_US if world > 1 else 0.0) for _, t in colls)
            ms = out[key]["ms_per_step"] + 1e-3 * wire_us
            lines.append({"metric": "Newton KKT solves/sec", "value": round(1e3 / ms, 2), "unit": "KKT solves/s", "n_gpus": world, "ms_per_step": round(ms, 3),
                          "scaling": "strong", "emulated": True, "top": top,
                          "how": "rank %d of %d emulated on ONE GPU through the production host code (kernels with that rank's grid sizes and data "
                                 "volumes, one-rank RCCL collectives) + modelled wire time" % (state["rank"], world),
                          "kernels_and_host_ms": out[key]["ms_per_step"], "wire_model_us": round(wire_us, 1),
                          "wire_model": {"link_GBps": LINK_GBS, "link_efficiency": LINK_EFF, "latency_us_per_collective": COLL_LAT_US,
                                         "bytes_received_per_rank": {"per_exchange_of_1_rhs": int(8 * width1 * (world - 1)),
                                                                     "Schur_exchange": int(8 * widthm * (world - 1))},
                                         "collectives": [{"what": w_, "transfer_us": round(t, 1)} for w_, t in colls]}})
        out["scale_lines_emulated"] = lines
        for ln in lines:
            print(json.dumps(ln), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/shard_step_emul.json", "w"), indent=1)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

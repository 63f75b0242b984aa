"""Owned / replicated time budget of the subtree-sharded step on ONE GPU.

For world = 2, 4, 8 the partition of rank 0 is installed and every phase of the sharded step is timed on its own
with events on the launch stream: the phases over set 1 are what shrinks with the number of ranks, the phases over
set 2 (the replicated top) + potrf / potrs are the Amdahl term.  The boundary exchange is left out (a full
single-rank factorisation beforehand leaves every clique's update block in the workspace, so the top sees valid
children); collectives are not timed here.

  python tools/shard_budget.py [--workload synth50k] [--reps 5]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_workload                      # noqa: E402
from smcp_amd import chordal, problems               # noqa: E402
from smcp_amd.cspmatrix import cspmatrix              # noqa: E402
from smcp_amd.kkt import KKTSystem                    # noqa: E402
from smcp_amd.shard import subtree_partition          # noqa: E402
from smcp_amd.symbolic import Symbolic                # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="synth50k")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--worlds", default="2,4,8")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    pat, m, density, label = build_workload(args.workload)
    symb = Symbolic(pat)
    cptr, cidx, cval = problems.random_constraints(symb, m, density=density, seed=1)
    kkt = KKTSystem(symb, cptr, cidx, cval, max_rhs=m)
    S = cspmatrix(symb, torch.from_numpy(problems.random_factor_blkval(symb, seed=0)).cuda())
    chordal.llt(S)
    msk = np.zeros(symb.blklen, dtype=bool)
    msk[symb.ccs_to_blk()] = True
    bx0 = torch.from_numpy(np.random.default_rng(2).standard_normal(symb.blklen) * msk).cuda()
    by0 = torch.from_numpy(np.random.default_rng(3).standard_normal(m)).cuda()

    def timed(f):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        f()
        b.record()
        b.synchronize()
        return a.elapsed_time(b)

    out = {}
    for world in [int(w) for w in args.worlds.split(",")]:
        P = subtree_partition(symb, max(world, 2))
        if world == 1:                                   # everything owned: the single-GPU step through the same phases
            P.owner[:] = 0
        kkt.partition = P
        kkt._apply_partition(P, 0)
        ranges = list(P.ranges_by_rank[0]) + list(P.top_ranges) if world > 1 else [(0, symb.blklen)]
        acc = {}
        for rep in range(args.reps + 1):
            t = {}
            L = S.copy()
            chordal.cholesky(L)                          # leaves every update block valid for the top passes below
            L.blkval.copy_(S.blkval)
            t["chol.own"] = timed(lambda: kkt._chol_part(L, 1))
            t["chol.top"] = timed(lambda: kkt._chol_part(L, 2))
            Y = L.copy()
            t["pinv.top"] = timed(lambda: kkt._pinv_part(Y, 2))
            t["pinv.own"] = timed(lambda: kkt._pinv_part(Y, 1))
            t["prep.top"] = timed(lambda: kkt._prepare_part(L, Y, 2))
            t["prep.own"] = timed(lambda: kkt._prepare_part(L, Y, 1))
            kkt.__dict__["_spair"] = (L, Y, (L.state(), Y.state()))
            t["gram.prep"] = timed(kkt._gram_prepare_part)
            t["gram.own"] = timed(lambda: kkt._gram_sweep(1, 0, m))
            t["gram.top"] = timed(lambda: kkt._gram_sweep(2, 0, m))
            t["gram.acc.own"] = timed(lambda: kkt._gram_accumulate(ranges))
            t["potrf.top"] = timed(kkt._potrf)
            bx = cspmatrix(symb, bx0.clone())
            by = by0.clone()
            for tag in ("h1", "h2"):
                t[tag + ".up.own"] = timed(lambda: kkt._hess_part(bx, 1, 0))
                t[tag + ".up.top"] = timed(lambda: kkt._hess_part(bx, 2, 0))
                t[tag + ".dn.top"] = timed(lambda: kkt._hess_part(bx, 2, 1))
                t[tag + ".dn.own"] = timed(lambda: kkt._hess_part(bx, 1, 1))
                if tag == "h1":
                    t["amap.own"] = timed(lambda: kkt.amap(bx))
                    t["potrs.top"] = timed(lambda: kkt._potrs(by))
                    t["aadj.own"] = timed(lambda: kkt.aadj(by))
            if rep:
                for k, v in t.items():
                    acc[k] = acc.get(k, 0.0) + v / args.reps
        own = sum(v for k, v in acc.items() if k.endswith(".own"))
        top = sum(v for k, v in acc.items() if not k.endswith(".own"))
        cl = np.asarray(P.owner)
        out[world] = dict(owned_ms=round(own, 3), replicated_ms=round(top, 3), top_cliques=int((cl == -1).sum()),
                          owned_cliques=int((cl == 0).sum()), exchange_roots=[len(r) for r in P.roots_by_rank],
                          phases={k: round(v, 3) for k, v in acc.items()})
        print("world %d: owned %.3f ms  replicated %.3f ms  (top cliques %d, owned by rank 0: %d)"
              % (world, own, top, out[world]["top_cliques"], out[world]["owned_cliques"]), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/r03_shard_budget.json", "w"), indent=1)
    print(json.dumps(out[max(out)]["phases"]))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Newton-KKT solve benchmark (BASELINE.json metric) on MI355X.

One "step" = one Newton-KKT solve at a new scaling point: cholesky(S) +
projected_inverse (dual scaling, solvers.py:881-891) -> Schur complement build over the m
constraints (solvers.py:479-497) -> potrf(H) (501) -> one solve_ call (506-541).
Inputs are synthetic (seeded numpy) and resident in HBM before the timed region.

N = 1 workload: the configuration the metric is quoted on, "synth50k" (config 5: nested
block-arrow chordal SDP, n = 50000, 8073 cliques, m = 100), which fits one GPU.
N > 1 (strong scaling of ONE solve): the clique tree is cut into subtrees owned by single ranks plus a
small replicated top (smcp_amd/shard.py).  EVERY sweep of the step is sharded by subtree -- cholesky,
projected_inverse, the Schur sweeps of all m constraints, both Hessians of solve_; the packed update blocks
of the subtree roots are exchanged over RCCL, the top is swept redundantly, each rank forms the partial
Gram matrix of its blkval ranges and one all-reduce of the m x m matrix H completes it; potrf / potrs of H
are replicated.  (--shard columns selects the simpler column sharding of H.)

Prints ONE JSON line on rank 0 (contract in the task statement) including `roofline` for the
dominant kernel (HIP-event timing inside the timed region), `cpu_baseline` (the CPU oracle timed on this
host, median of three repeats), `back_solve` (the solve_ sub-rate of SURVEY 8d: two Hessians + potrs on
the factored system) and, for the default N = 1 run, `secondary`: the GPU-only figures of BASELINE.json's
configs 2, 3 and 4 (dense4096, arrow, maxcut; a few steps each, no CPU leg; --no-secondary skips them).
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PMC_FILE = "r05_pmc_%s.json"       # per workload key: tools/pmc_step.sh (three rocprofv3 --pmc passes; one step = one period of the launch sequence)
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)
FP64_PEAK_TFLOPS = 78.6   # vendor fp64 vector = matrix peak


def build_workload(name, seed=0):
    from smcp_amd import problems
    if name == "synth50k":
        pat = problems.nested_block_arrow_pattern(seed=seed)
        m, density, label = 100, 0.005, "synth50k nested block-arrow SDP n=50000, 8073 cliques, m=100"
    elif name == "synth50k_dense":
        # the headline pattern with ten times denser constraints (0.05 |V| entries each: ~120 per (family, constraint), beyond the
        # 48-entry gate of the entry-driven family sweep k_fam_terms): what the step costs when k_fam_sparse / k_hess_up_fam take over
        pat = problems.nested_block_arrow_pattern(seed=seed)
        m, density, label = 100, 0.05, "synth50k pattern, m=100 constraints of density 0.05 (family sweep beyond the k_fam_terms gate)"
    elif name == "synth50k_trace":
        # the headline problem with ONE constraint replaced by a diagonal matrix (the trace constraint of an SDP relaxation): 55 entries
        # in every (family, constraint) list of that constraint, 13 in the others.  Up to round 4 the longest list chose the route for
        # the whole set (beyond 48: the dense family sweep, as synth50k_dense); since round 5 the mean does and long lists go in chunks
        pat = problems.nested_block_arrow_pattern(seed=seed)
        m, density, label = 100, 0.005, "synth50k with one diagonal (trace) constraint among the 100: one long (family, constraint) list among short ones"
    elif name == "band200":
        # config 1's problem as a KKT-solve point: band pattern n = 200, half-bandwidth 3, m = 100 constraints DENSE on V (band_SDP,
        # base.py:617-632) -- the far end of the density range on a pattern of 197 cliques of four columns: launch-bound
        pat = problems.band_pattern(200, 3)
        m, density, label = 100, None, "band SDP n=200, half-bandwidth 3, m=100 constraints dense on V (config 1's problem)"
    elif name == "synth6k":   # reduced copy for quick checks only (NOT the benchmark)
        pat = problems.nested_block_arrow_pattern(nsub=2, nmid=56, seed=seed)
        m, density, label = 100, 0.005, "synth6k (reduced, check only)"
    elif name == "arrow":
        pat = problems.block_arrow_pattern(2000, 64, 128)
        m, density, label = 100, 0.005, "block-arrow 2000x64+128, m=100"
    elif name == "dense4096":
        pat = problems.band_pattern(4096, 4095)
        m, density, label = 16, 0.005, "single dense clique n=4096, m=16"
    elif name == "maxcut":
        # config 4: max-cut relaxation on a random 1000-node / 5909-edge graph (the size of SDPLIB maxG51), m = n
        # constraints A_i = e_i e_i^T -- all column-sparse, so the Schur complement takes the trsm + SCMcolumn2 route
        # (solvers.py:489-497, misc.c:620-663).  Pattern, embedding and constraints exactly as the drivers build them.
        return None, 1000, None, "max-cut SDP, random graph n=1000, 5909 edges (maxG51 size), m=1000, SCMcolumn2 path"
    else:
        raise SystemExit("unknown workload " + name)
    return pat, m, density, label


def maxcut_problem():
    """(symbolic, cptr, cidx, cval) of the max-cut workload through the drivers' own index algebra (solvers._Problem)."""
    from smcp_amd import base, solvers
    P = base.maxcut_SDP(1000, 5909, seed=0)
    pr = solvers._Problem(P._A, P._b)
    cptr, cidx, cval = pr._con
    return pr.symb, cptr, cidx, cval


def lds_fits(nn, na):
    """Mirror of mfma_lds_doubles() in csrc/front_mfma.hip: does the working set of a front fit 160 KB of LDS?"""
    pad = lambda x: x | 1
    lk, ll, lf = pad(na), pad(nn), pad(nn + na)
    d = lk * nn + ll * nn + lk * na + lf * nn + ll * nn + lk * nn + lk * nn + ll * nn + lk * na + 256 + 8
    return d * 8 <= 160 * 1024 - 2048


def self_launch(ngpus):
    """python bench.py --gpus N without a launcher: N ranks through torch.distributed.run as a child process."""
    import socket
    import subprocess
    import torch           # device_count() does not initialise the GPU on this image
    ndev = torch.cuda.device_count()
    if os.environ.get("SMCP_BENCH_BACKEND", "nccl") == "nccl" and ndev < ngpus:
        print("bench.py: --gpus %d needs %d visible GPUs for one RCCL rank per GPU, found %d"
              % (ngpus, ngpus, ndev), file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def git_sha():
    try:
        import subprocess
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        return None


def csrc_sha256():
    """sha256 over the kernel sources (smcp_amd/csrc/*, names and contents, sorted) -- the GPU box has no .git, so this is
    what ties a committed PMC summary to the code that is running."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "smcp_amd", "csrc")
    for name in sorted(os.listdir(d)):
        h.update(name.encode())
        with open(os.path.join(d, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def famt_executed_flops(symb, cptr, cidx, chunks, tiles="all"):
    """Flops k_fam_terms EXECUTES per sweep of all chunks (front_famt.hip): one wave per (family parent, right-hand side)
    issues ks = ceil(2 T / 4) steps of NAT (NAT + 1) / 2 + NAT + 1 v_mfma_f64_16x16x4 (2048 flop each, tile padding
    included), T = entries of the constraint inside the family (the parent's and its children's; up to 384, in chunks of 48 / 64 entries).  This is
    what SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 counts (profiles/r04_mfma_pmc.txt)."""
    fam = symb.family_roles()
    nn_, na_ = symb.clique_sizes()
    par = np.asarray(symb.snpar)
    blkptr = np.asarray(symb.blkptr)
    m = len(cptr) - 1
    parents = np.flatnonzero(fam == 2)
    if not len(parents):
        return 0.0
    pid = -np.ones(symb.Nsn, dtype=np.int64)
    pid[parents] = np.arange(len(parents))
    owner = np.where(fam == 2, pid, np.where(fam == 1, pid[np.maximum(par, 0)], -1))
    clique = np.searchsorted(blkptr, np.asarray(cidx), side="right") - 1
    con = np.repeat(np.arange(m), np.diff(cptr))
    o = owner[clique]
    T = np.zeros((len(parents), m), dtype=np.int64)
    np.add.at(T, (o[o >= 0], con[o >= 0]), 1)
    T = np.minimum(T, 384)                             # FAMT_TMAX; lists go through in chunks: 48 entries of descriptors in k_fam_terms,
    chunk = 64 if tiles == "updates" else 48           # 64 terms (one per lane) in the fused extend-add -- steps are padded per chunk
    full, rest = T // chunk, T % chunk
    ks = full * ((2 * chunk + 3) >> 2) + ((2 * rest + 3) >> 2)
    nat = (int(na_[parents].max()) + 15) // 16          # the launch's instantiation serves the widest parent
    # tiles per step: update tiles nat (nat + 1) / 2, Q tiles nat, G_NN 1 -- all in k_fam_terms<NAT, true>; with the fused
    # extend-add k_fam_terms<NAT, false> issues the Q and G_NN tiles and k_lf_assemble_fz the update tiles
    per_step = {"all": nat * (nat + 1) // 2 + nat + 1, "panels": nat + 1, "updates": nat * (nat + 1) // 2, "useful_updates": 0}[tiles]
    nrhs = sum(chunks)
    if tiles == "useful_updates":
        # what the update tiles are FOR: per entry (term) a symmetric rank-2 update s (a(x) a(y)^T + a(y) a(x)^T) of the parent's
        # na x na update matrix, lower triangle only -- 2 multiply-adds per entry, na (na + 1) / 2 entries: no tile padding (na
        # rounded up to 16, the full diagonal tiles), no padding of the term count to a multiple of two
        nap = na_[parents].astype(np.float64)
        return float((T[:, :nrhs] * (2.0 * nap * (nap + 1.0))[:, None]).sum())
    return float(ks[:, :nrhs].sum()) * per_step * 2048.0


def pmc_file(workload):
    """The committed PMC summary of a workload (tools/pmc_step.sh) -- only when the kernel sources are byte-identical to the ones
    the counters were collected on (csrc_sha256); otherwise (None, why)."""
    name = PMC_FILE % workload
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None, "no profiles/%s" % name
    if tj.get("csrc_sha256") != csrc_sha256():
        return None, "dropped: profiles/%s was collected on other kernel sources (csrc_sha256 %s, running %s)" % (name, tj.get("csrc_sha256"), csrc_sha256())
    return tj, "profiles/%s (rocprofv3 --pmc passes, csrc_sha256 %s = the running sources)" % (name, tj.get("csrc_sha256"))


def pmc_traffic(dom, workload, world):
    """HBM bytes per launch of kernel `dom` from the committed PMC summary (separate rocprofv3 --pmc passes, FETCH_SIZE doubled as
    the gfx950 guide prescribes), or (None, why)."""
    if world != 1:
        return None, None
    tj, src = pmc_file(workload)
    if tj is None:
        return None, src
    hits = [v for kname, v in tj.get("kernels", {}).items() if kname.split("<")[0] == dom.split("<")[0]]
    if not hits:
        return None, "profiles/%s does not list %s" % (PMC_FILE % workload, dom)
    return max(v["hbm_bytes_per_launch"] for v in hits), src


def step_roofline(workload, ms_per_step, world):
    """The WHOLE step against both roofs: counter traffic of one step / its duration against the HBM peak,
    matrix-pipe flops executed in one step / its duration against the fp64 MFMA peak -- both sums from the committed PMC
    summary of this workload, the duration from this run."""
    if world != 1:
        return None
    tj, src = pmc_file(workload)
    if tj is None:
        return {"source": src}
    b, f = float(tj["step"]["hbm_bytes"]), float(tj["step"]["mfma_flops"])
    gbs, tfl = b / (1e-3 * ms_per_step) / 1e9, f / (1e-3 * ms_per_step) / 1e12
    return {"hbm": {"bytes_per_step": int(b), "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)},
            "mfma": {"executed_flops_per_step": int(f), "achieved": round(tfl, 2), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(tfl / FP64_PEAK_TFLOPS, 4)},
            "launches_per_step": tj.get("launches_per_step"), "source": src}


def kernel_roofline(dom, dom_ms, dom_launches, breakdown, symb, m, max_rhs, mloc, part, rank, world, workload, con=None):
    """roofline object of the dominant kernel `dom` (name as rocprofv3 shows it, without template arguments):
    achieved = ALGORITHMIC bytes (or flops) of its launches in one step / their duration (HIP events).
    B = sum (nn + na) nn (blkval), U = sum na^2, Up = sum na (na + 1) / 2 (packed update blocks)."""
    nn_, na_ = symb.clique_sizes()
    lds_ok = np.array([lds_fits(int(a), int(b)) for a, b in zip(nn_, na_)])
    chunks = [min(max_rhs, mloc - c) for c in range(0, mloc, max_rhs)]
    if part is not None:     # this rank sweeps only its own subtrees and the replicated top
        lds_ok = lds_ok & ((part.owner == rank) | (part.owner == -1))
        chunks = [min(max_rhs, m - c) for c in range(0, m, max_rhs)]
    owned = np.ones(symb.Nsn, dtype=bool) if part is None else ((part.owner == rank) | (part.owner == -1))
    Bk = ((nn_ + na_) * nn_).astype(np.float64)
    Uk = (na_ * na_).astype(np.float64)
    Upk = (na_ * (na_ + 1) // 2).astype(np.float64)
    par = symb.snpar
    Uch = np.zeros(symb.Nsn)
    np.add.at(Uch, par[par >= 0], Uk[par >= 0])       # update volume a clique reads from its children
    Upch = np.zeros(symb.Nsn)
    np.add.at(Upch, par[par >= 0], Upk[par >= 0])
    nch = np.diff(symb.chptr)
    fam = symb.family_roles()
    fam_mask = lds_ok & (fam > 0)
    leaf_gram = "k_leaf_pairs" in breakdown           # the family children's panels are not formed (front_leafgram.hip)
    B = Bk[owned].sum()
    U = Uk[owned].sum()
    solve_chunks = chunks + [1, 1]                    # the two single right-hand sides of solve_

    def cls_bytes(mask):     # two-directional sweep kernels: panels read + written, updates written + read
        return sum(8.0 * (r * (2 * Bk[mask].sum() + Uk[mask].sum() + Uch[mask].sum()) + Bk[mask].sum()) for r in solve_chunks)

    def up_bytes(mask):      # Schur sweeps build their input from the entry lists (panels written only); solve_'s read them too
        tot = 0.0
        for i, r in enumerate(solve_chunks):
            panel = (1 if i < len(chunks) else 2) * Bk[mask].sum()
            tot += 8.0 * (r * (panel + Upk[mask].sum() + Upch[mask].sum()) + Bk[mask].sum())
        return tot

    note = None
    extra = {}
    alg = None
    fused = "k_lf_assemble_fz" in breakdown           # the parents' updates are formed by the extend-add above (front_famt.hip)
    if dom == "k_lf_assemble_fz" and con is not None and part is None:
        # Fused extend-add: per launch it HAS to write the assembled fronts (panel + update block of every (front, right-hand side))
        # and read the families' tables and the static term lists once; the operands it gathers per term come from L2 / MALL
        # (~2.2 GB per launch on synth50k, not HBM traffic by construction).  Executed matrix-core work: the update tiles.
        par2 = fam == 2
        big = np.zeros(symb.Nsn, dtype=bool)
        big[np.unique(np.asarray(par)[par2])] = True      # the fronts this launch assembles: the parents of the family parents
        nnb, nab = nn_[big].astype(np.float64), na_[big].astype(np.float64)
        fronts = 8.0 * sum(chunks) * float(((nnb + nab) * nnb + nab * nab).sum())
        natz = (int(na_[par2].max()) + 15) // 16
        cnn = int(nn_[fam == 1].max()) if (fam == 1).any() else 1
        rec = 256 + 2 * 16 * natz * 16 + (16 * natz) ** 2 + 16 * natz + 8 * cnn * (16 + 2 * 16 * natz)
        tables = 8.0 * float(par2.sum()) * rec + 12.0 * float(len(con[1]))
        alg = fronts + tables
        ex = famt_executed_flops(symb, con[0], con[1], chunks, "updates")
        avg_s = 1e-3 * dom_ms / dom_launches
        ai = ex / alg
        ridge = FP64_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
        tfl = ex / dom_launches / avg_s / 1e12
        gbs = alg / dom_launches / avg_s / 1e9
        mf = {"achieved": round(tfl, 2), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tfl / FP64_PEAK_TFLOPS, 4)}
        hb = {"achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)}
        top = mf if ai >= ridge else hb
        traffic, source = pmc_traffic(dom, workload, world)
        useful = famt_executed_flops(symb, con[0], con[1], chunks, "useful_updates")
        # (neither pipe is half busy: what bounds the launch is the dependent chain of a task -- gather the operands of a term group
        # from LDS, 10 MFMAs, the next group -- on one workgroup per CU, not a roof: said so, with the roof the arithmetic intensity
        # points at kept beside it)
        latency = max(mf["frac"], hb["frac"]) < 0.5
        return {"kernel": dom, "bound": "latency" if latency else ("mfma" if ai >= ridge else "hbm"), "bound_by_intensity": "mfma" if ai >= ridge else "hbm",
                "achieved": top["achieved"], "peak": top["peak"], "unit": top["unit"],
                "frac": top["frac"], "traffic": traffic, "traffic_source": source,
                "traffic_over_algorithmic": round(traffic / (alg / dom_launches), 2) if traffic else None,
                "mfma": dict(mf, flops_per_launch=ex / dom_launches, useful_flops_per_launch=useful / dom_launches,
                             useful_share=round(useful / ex, 3) if ex else None),
                "hbm": dict(hb, bytes_per_launch=alg / dom_launches), "arithmetic_intensity_flop_per_byte": round(ai, 2),
                "ridge_flop_per_byte": round(ridge, 2), "avg_launch_us": round(1e6 * avg_s, 2), "launches_per_step": dom_launches,
                "note": "fused extend-add (k_lf_assemble_fz): algorithmic bytes = assembled fronts written + family tables and term lists "
                        "read once; executed flops = update tiles (v_mfma_f64_16x16x4 x 2048) counted on the host; the bound is derived "
                        "from their ratio against the ridge.  The packed updates of the family parents (1.49 GB written + 1.49 GB read per "
                        "launch pair in round 3) no longer exist."}
    if dom in ("k_fam_sparse", "k_fam_terms"):
        # bytes the fused family kernel HAS to move: the parents' output panels (+ the children's when they are formed),
        # the parents' packed updates, the constants of every member once per launch; the children's updates stay in LDS
        panels = Bk[fam_mask & (fam == 2)].sum() + (0.0 if leaf_gram else Bk[fam_mask & (fam == 1)].sum())
        upd_out = 0.0 if fused else Upk[fam_mask & (fam == 2)].sum()      # fused extend-add: the packed updates are never written
        alg = sum(8.0 * (r * (panels + upd_out) + Bk[fam_mask].sum()) for r in chunks)
        nnk, nak = nn_[fam_mask].astype(np.float64), na_[fam_mask].astype(np.float64)
        flops = float((nnk ** 3 + 3 * nak * nnk ** 2 + 3 * nak ** 2 * nnk).sum()) * sum(chunks)
        per_level = sum(8.0 * (r * (Bk[fam_mask].sum() + Upk[fam_mask].sum() + Upch[fam_mask].sum()) + Bk[fam_mask].sum()) for r in chunks)
        extra = {"per_level_bytes_per_launch": per_level / dom_launches,
                 "mfma_canonical": {"flops_per_launch": flops / dom_launches,
                                    "achieved_tflops": round(flops / (1e-3 * dom_ms) / 1e12, 2),
                                    "frac": round(flops / (1e-3 * dom_ms) / 1e12 / FP64_PEAK_TFLOPS, 4),
                                    "note": "dense-formulation flops of the same sweeps (SURVEY 8d); NOT executed by this kernel"}}
        if dom == "k_fam_terms" and con is not None and part is None:
            ex = famt_executed_flops(symb, con[0], con[1], chunks, "panels" if fused else "all")
            extra["mfma"] = {"flops_per_launch": ex / dom_launches,
                             "achieved_tflops": round(ex / (1e-3 * dom_ms) / 1e12, 2),
                             "frac": round(ex / (1e-3 * dom_ms) / 1e12 / FP64_PEAK_TFLOPS, 4),
                             "note": "EXECUTED v_mfma_f64_16x16x4 x 2048 flop, counted on the host from the entries per (family, "
                                     "constraint); equals SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 of the PMC pass"}
        if dom == "k_fam_terms":      # tables of the family (record of k_famt_prep) read once per workgroup instead of the constants
            alg = sum(8.0 * r * (panels + upd_out) for r in chunks) + 8.0 * Bk[fam_mask].sum()
        note = ("bytes = the parents' output panels%s + the parents' packed updates + the members' constants (the children's "
                "update matrices never exist); per_level_bytes = SURVEY 8d's per-level figure for the same sweeps; mfma = "
                "executed matrix-core flops, mfma_canonical = dense-formulation flops" % (" (the children's panels are not formed: their Gram block comes from "
                                                       "k_leaf_pairs in closed form)" if leaf_gram else " + the children's"))
    elif dom == "k_hess_up_fam":
        alg = sum(8.0 * (r * (Bk[fam_mask].sum() + Upk[fam_mask & (fam == 2)].sum()) + Bk[fam_mask].sum()) for r in chunks)
    elif dom == "k_gram_diag128":
        rows = B - (Bk[owned & (fam == 1)].sum() if leaf_gram else 0.0)
        alg = 8.0 * (m * rows + rows)                 # one read of the swept stack (+ the weights)
        note = "bytes = one read of the swept constraint stack (m x rows) + the weights"
    elif dom in ("k_lf_assemble_lds", "k_lf_assemble_lds_dyn"):
        big = (~lds_ok) & owned & (nch > 0)
        nf_ = (nn_ + na_).astype(np.float64)
        alg = sum(8.0 * r * (Upch[big].sum() + (nf_[big] * (nf_[big] + 1) / 2).sum()) for r in solve_chunks)
        note = "bytes = the children's packed updates read once + the assembled lower triangle of every front written once, per right-hand side"
    elif dom == "k_lfsp_up":
        sp_mask = (~lds_ok) & (nch == 0) & (nn_ <= 64) & (na_ <= 128) & (na_ > 0)
        alg = sum(8.0 * (r * (Bk[sp_mask].sum() + Upk[sp_mask].sum()) + 3 * (2 * Bk[sp_mask].sum() + Uk[sp_mask].sum())) for r in chunks)
        note = ("three launches per chunk of right-hand sides (Q, update, G_NN) are timed under this name; bytes = output panels + "
                "packed updates of every (front, constraint) + the fronts' constants once per launch, averaged over the launches")
    elif dom in ("k_hess_up_level", "k_hess_down_level"):
        alg = sum(8.0 * (r * (2 * B + 2 * U) + B) for r in solve_chunks)
    elif dom == "k_hess_up_pad":
        alg = up_bytes(lds_ok)
    elif dom == "k_hess_up_n16":
        alg = up_bytes(lds_ok & (nn_ <= 16) & (na_ <= 64))
    elif dom in ("k_hess_up_mfma<true>", "k_hess_down_mfma<true>"):
        alg = cls_bytes(lds_ok)
    elif dom in ("k_hess_up_mfma<false>", "k_hess_down_mfma<false>"):
        alg = cls_bytes(~lds_ok)
    elif dom in ("k_chol_level", "k_pinv_level"):
        alg = 8.0 * (2 * B + 2 * U)
    elif dom in ("k_trsm_fwd_level", "k_trsm_bwd_level", "k_trsm_fwd_mfma", "k_trsm_bwd_mfma"):
        # supernodal triangular solves with the n x |K| right-hand sides of the SCMcolumn2 route: the factor is read once
        # per launch chain, every right-hand-side column is read and written once per sweep
        ncols = float(m)       # one column of S^-1 per max-cut constraint
        alg = 8.0 * (B + 2.0 * symb.n * ncols)
        note = "bytes = the factor once + the n x m right-hand sides read and written once per sweep direction"
    if alg is not None:
        avg_s = 1e-3 * dom_ms / dom_launches
        per_launch = alg / dom_launches
        achieved = per_launch / avg_s / 1e9
        traffic, source = pmc_traffic(dom, workload, world)
        roofline = {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": source,
                    "bytes_per_launch": per_launch, "avg_launch_us": round(1e6 * avg_s, 2),
                    "launches_per_step": dom_launches}
        roofline.update(extra)
        if "mfma" in roofline:     # the bound is derived, not asserted: arithmetic intensity of the executed work against the ridge
            ai = roofline["mfma"]["flops_per_launch"] / per_launch
            ridge = FP64_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
            roofline["arithmetic_intensity_flop_per_byte"] = round(ai, 2)
            roofline["ridge_flop_per_byte"] = round(ridge, 2)
            roofline["bound"] = "hbm" if ai < ridge else "mfma"
        if note:
            roofline["note"] = note
        return roofline
    # large-front phase kernels (configs 2 and 3): MFMA-bound, canonical BLAS-3 flop counts per (front, rhs)
    # (SURVEY 8d): up1 = E, T (2 na nn^2 + nn^3); up2 = update, G, G_NN (2 na^2 nn + na nn^2 + nn^3 / 3);
    # up3 = scaling product (na^2 nn, triangular operand)
    big = ~lds_ok
    nnf, naf = nn_[big].astype(np.float64), na_[big].astype(np.float64)
    # fronts of more than six 64-column tiles form G_NN = Li F_NN Li^T as Z = Li Fl (nn^3 / 3, phase 1) and
    # Z Li^T + Li Z^T (2 nn^3 / 3, phase 2) instead of T = Li F_NN (nn^3) and T Li^T (nn^3 / 3): front_large.hip lf_sym_split
    split = nnf > 6 * 64
    lf_flops = {"k_lf_up1": (2 * naf * nnf ** 2 + np.where(split, nnf ** 3 / 3, nnf ** 3)).sum(),
                "k_lf_up2": (2 * naf ** 2 * nnf + naf * nnf ** 2 + np.where(split, 2 * nnf ** 3 / 3, nnf ** 3 / 3)).sum(),
                "k_lf_up3": (naf ** 2 * nnf).sum()}
    if dom in lf_flops:
        nr = sum(solve_chunks)
        tfl = lf_flops[dom] * nr / (1e-3 * dom_ms) / 1e12
        traffic, source = pmc_traffic(dom, workload, world)
        return {"kernel": dom, "bound": "mfma", "achieved": round(tfl, 2), "peak": FP64_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(tfl / FP64_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": source,
                "flops_per_step": lf_flops[dom] * nr, "avg_launch_us": round(1e3 * dom_ms / dom_launches, 2),
                "launches_per_step": dom_launches}
    if dom == "k_chol_flow":
        # one-launch blocked Cholesky (front_flow.hip): per step the single fronts beyond 272 rows without separator (roots), the
        # Y_AA blocks of single fronts with a separator beyond 272 rows, and H (m > 128).  n^3 / 3 flops each; what bounds it is the
        # chain of 64-column diagonal blocks (factor + invert + hand-over, ~35 us each), not a roof.
        nf_ = nn_ + na_
        sizes = [float(v) for v in nn_[big & (nf_ > 272) & (na_ == 0)]] + [float(v) for v in na_[big & (na_ > 272)]] + ([float(m)] if m > 128 else [])
        fl_step = sum(v ** 3 / 3.0 for v in sizes)
        blocks = sum(math.ceil(v / 64.0) for v in sizes)
        tfl = fl_step / (1e-3 * dom_ms) / 1e12
        traffic, source = pmc_traffic(dom, workload, world)
        return {"kernel": dom, "bound": "latency", "bound_by_intensity": "mfma", "achieved": round(tfl, 4), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(tfl / FP64_PEAK_TFLOPS, 5), "traffic": traffic, "traffic_source": source, "flops_per_step": fl_step,
                "matrix_orders_upper_bound": sizes, "diagonal_blocks_per_step": blocks,
                "us_per_diagonal_block": round(1e3 * dom_ms / max(blocks, 1), 2),
                "avg_launch_us": round(1e3 * dom_ms / dom_launches, 2), "launches_per_step": dom_launches,
                "note": "latency chain: the 64-column diagonal blocks of a factorisation are factored one after the other (potrf + inverse in "
                        "LDS, four dependent 16 x 16 steps, then the hand-over to the next block's workgroup); panel and trailing tiles run "
                        "beside the chain inside the same launch.  The quantity to reduce is us_per_diagonal_block."}
    if dom == "k_lf_diag":
        # the diagonal-block step of the blocked Cholesky factorisations (config 4: the root's front, chol(Y_AA) of the top
        # fronts, potrf(H)): per 64-wide block one workgroup factors the block and inverts its factor, 2 * 64^3 / 3 flops,
        # as a dependent chain of four 16 x 16 steps.  Fronts of at most 272 rows take k_mid_chol instead (front_large.hip).
        nf_ = nn_ + na_
        blocks = float(np.ceil(nn_[big & (nf_ > 272)] / 64.0).sum() + np.ceil(na_[big & (na_ > 272)] / 64.0).sum() +
                       (math.ceil(m / 64.0) if m > 128 else 0))
        fl_step = blocks * 2.0 * 64.0 ** 3 / 3.0
        tfl = fl_step / (1e-3 * dom_ms) / 1e12
        return {"kernel": dom, "bound": "latency", "bound_by_intensity": "mfma", "achieved": round(tfl, 5), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(tfl / FP64_PEAK_TFLOPS, 6), "traffic": None, "flops_per_step": fl_step, "diagonal_blocks_per_step": blocks,
                "avg_launch_us": round(1e3 * dom_ms / dom_launches, 2), "launches_per_step": dom_launches,
                "note": "latency chain, not a throughput kernel: one workgroup per front factors and inverts ONE 64 x 64 diagonal block per "
                        "launch (four dependent 16 x 16 steps, ~31 us); the fraction says how far such a chain is from the matrix peak, "
                        "the quantity to reduce is launches_per_step x avg_launch_us"}
    return {"kernel": dom, "bound": None, "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
            "avg_launch_us": round(1e3 * dom_ms / dom_launches, 2), "launches_per_step": dom_launches,
            "note": "no byte / flop model for this kernel in bench.py"}


def run_workload(args, workload, steps, warmup, want_cpu, primary, env):
    """One workload through the timed protocol; returns the result dict on rank 0 (None elsewhere).
    want_cpu: False / None = no CPU leg, "full" = the oracle on the whole unit, median of --cpu-repeats (headline), "quick" = one
    repeat on the host BLAS with one Schur column per thread, scaled (the secondary legs: a few seconds each)."""
    if want_cpu is True:
        want_cpu = "full"
    import torch
    import torch.distributed as dist
    from smcp_amd import chordal, problems
    from smcp_amd.cspmatrix import cspmatrix
    from smcp_amd.kkt import KKTSystem
    from smcp_amd.symbolic import Symbolic
    world, rank, dev, lib, force_sharded = env["world"], env["rank"], env["dev"], env["lib"], env["force_sharded"]

    # ---------------- problem (untimed) ----------------
    pat, m, density, label = build_workload(workload)
    if args.m and primary:
        m = args.m
    t0 = time.time()
    if workload == "maxcut":
        symb, cptr, cidx, cval = maxcut_problem()
    else:
        symb = Symbolic(pat)
        if workload == "band200":
            # as the drivers do (solvers._Problem, options['amalgamate']): a chain of 197 one-column cliques is 197 levels of
            # launches (15.9 ms per step, 63 solves/s against 1000 on the CPU); merged into cliques of up to 16 columns it is 13
            from smcp_amd.symbolic import amalgamate
            emb = amalgamate(symb)
            if emb is not None:
                symb = Symbolic(emb[0], emb[1])
    t_sym = time.time() - t0
    fl = symb.flops()
    B, U = fl["B"], fl["U"]
    per_rhs = 8 * (U + 3 * B)
    max_rhs = (args.max_rhs if primary else None) or int(max(1, min(m, (48 << 30) // per_rhs)))
    if workload != "maxcut":
        cptr, cidx, cval = (problems.random_constraints(symb, m, density=density, seed=1) if density is not None
                            else problems.random_constraints(symb, m, seed=1, dense_on_v=True))
    if workload == "synth50k_trace":
        ccp, cri = symb.sparsity_pattern()
        dpos = np.sort(symb.ccs_to_blk()[ccp[:-1]]).astype(np.int64)        # the diagonal entries (first of every column of V)
        dval = np.random.default_rng(5).standard_normal(len(dpos))
        cols = [(cidx[cptr[j]:cptr[j + 1]], cval[cptr[j]:cptr[j + 1]]) for j in range(m)]
        cols[0] = (dpos, dval)
        cptr = np.concatenate([[0], np.cumsum([len(p_) for p_, _ in cols])]).astype(np.int64)
        cidx = np.concatenate([p_ for p_, _ in cols]).astype(np.int64)
        cval = np.concatenate([v_ for _, v_ in cols])
    # the reference's default classification (tnzcols = 0.1): on the synthetic patterns every constraint touches more
    # than n / 10 columns and is swept; the max-cut constraints are column-sparse
    kkt = KKTSystem(symb, cptr, cidx, cval, max_rhs=max_rhs, tnzcols=0.0 if args.kktsolver == "qr" else None)
    placement = None
    part = None
    # max-cut: every constraint is column-sparse (SCMcolumn2 route) -- with N > 1 the factors are replicated (n = 1000: the
    # factorisation is not what costs) and the constraints sharded over the ranks (kkt_schur_gram_part), one all-reduce of H
    shard = "columns" if workload == "maxcut" else args.shard
    if (world > 1 or force_sharded) and shard == "subtree":
        kkt.force_sharded = force_sharded
        part = kkt.set_partition(dist.group.WORLD)   # subtrees -> ranks, replicated top, boundary exchange lists
    Lh = problems.random_factor_blkval(symb, seed=0)
    S = cspmatrix(symb, torch.from_numpy(Lh).to(dev))
    chordal.llt(S)                       # S = L0 L0^T on V: positive definite by construction
    msk = np.zeros(symb.blklen, dtype=bool)
    msk[symb.ccs_to_blk()] = True
    rng = np.random.default_rng(2)
    bx0 = torch.from_numpy(rng.standard_normal(symb.blklen) * msk).to(dev)
    by0 = torch.from_numpy(rng.standard_normal(m)).to(dev)
    L = S.copy()
    Y = S.copy()
    bx = cspmatrix(symb, bx0.clone())
    by = by0.clone()
    H = kkt.H
    j0 = (m * rank) // world
    j1 = (m * (rank + 1)) // world
    st = lambda: torch.cuda.current_stream().cuda_stream
    h = symb.handle

    def chk(rc, what):
        if rc != 0:
            raise RuntimeError("%s failed rc=%d" % (what, rc))

    # failure reports deferred: the factorisations latch "not positive definite" on the device and the step reads the
    # latch once, at its end (chordal.lazy_status; SMCP_BENCH_EAGER=1: a read-back after every factorisation)
    lazy = os.environ.get("SMCP_BENCH_EAGER") != "1"
    split_calls = os.environ.get("SMCP_BENCH_SPLIT") == "1"
    if lazy:
        chordal.lazy_status(symb, True)

    def step():
        _step()
        if lazy:
            chordal.check_status(symb)

    def _step():
        bx.blkval.copy_(bx0)
        by.copy_(by0)
        if part is not None and args.kktsolver != "qr":
            # N > 1: every sweep of the step sharded by subtree (cholesky, projected_inverse, Schur sweeps, the two
            # Hessians of solve_); boundary update blocks + H + Amap + the completed x travel over RCCL
            Ls, Ys = kkt.factor_scaling(S, dist.group.WORLD, defer_status=True)   # status agreed with H's all-reduce
            # x stays sharded (valid on this rank's cliques and the top: what a sharded consumer reads); SMCP_BENCH_COMPLETE_X=1
            # adds the all-gather that completes it on every rank
            kkt.factor(Ls, Ys, dist.group.WORLD)(bx, by, 1.0, complete=os.environ.get("SMCP_BENCH_COMPLETE_X") == "1")
            return
        L.blkval.copy_(S.blkval)
        if split_calls:                          # SMCP_BENCH_SPLIT=1: the reference's call sequence, one library call each
            chordal.cholesky(L)                  # csp_cholesky
            Y.blkval.copy_(L.blkval)
            chordal.projected_inverse(Y)         # csp_projected_inverse (leaves the inverse-form factor of L for the sweeps)
        else:
            # the dual scaling point in one call (solvers.py:881-891): cholesky + projected_inverse + the separator factors
            # the sweeps need, their independent stages side by side (csp_cholesky_projected_inverse)
            chordal.cholesky_projected_inverse(L, Y)
        if args.kktsolver == "qr":
            kkt.factor_qr(L, Y, dist.group.WORLD if world > 1 else None)(bx, by, 1.0)
            return
        if split_calls or world > 1:
            # Schur complement: this rank's columns + one RCCL all-reduce (smcp_amd.kkt.ShardedSchur), then potrf
            kkt.build_schur(L, Y, dist.group.WORLD if world > 1 else None)
            kkt._potrf()
            chk(lib.kkt_solve(h, L.blkval.data_ptr(), Y.blkval.data_ptr(), H.data_ptr(), m, 1.0,
                              bx.blkval.data_ptr(), by.data_ptr(), st()), "solve")
        else:
            # f = kktsolver(L, Y); f(bx, by) (solvers.py:893, 506-541): kkt_schur_factor + kkt_solve -- with the status deferred
            # potrf(H) runs beside the first Hessian sweep of solve_
            kkt.factor(L, Y)(bx, by, 1.0)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_loop(nsteps):
        """exactly nsteps steps between barrier + synchronize on both sides; the MAX over the ranks, seconds"""
        barrier()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            step()
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    for _ in range(warmup):
        step()
    barrier()
    value_as_allocated = None
    if getattr(args, "tune_placement", 0) > 0 and workload == "synth50k" and primary:      # (every rank of an N-rank job tunes its own buffers)
        # The same K steps on the buffers AS ALLOCATED first (what a caller who does not tune gets), then
        # csp_tune(CSP_TUNE_PLACEMENT) -- untimed set-up like the symbolic analysis: the two output buffers of the family sweep go
        # to the fastest of up to N fresh allocations for its store pattern (the same kernel takes 0.74 or
        # 0.93 ms depending on where they lie) -- and the measurement proper.  --tune-placement 0: as allocated only.
        value_as_allocated = steps / timed_loop(steps)
        if lazy:
            chordal.check_status(symb)
        chordal.tune(symb, chordal.TUNE_PLACEMENT, int(args.tune_placement))
        rep = (ctypes.c_double * 3)()
        lib.csp_tune_report(symb.handle, rep)
        placement = {"tries": int(args.tune_placement), "probe_ms_before": round(rep[0], 4), "probe_ms_after": round(rep[1], 4)}
        for _ in range(max(1, min(warmup, 2))):
            step()
        barrier()
    prof = not args.no_profile
    back_solve = None
    nk = int(lib.csp_profile_kinds())
    names = [lib.csp_profile_kernel_name(i).decode() for i in range(nk)]

    def read_profile(nsteps):
        ms = (ctypes.c_double * nk)()
        cnt = (ctypes.c_int64 * nk)()
        lib.csp_profile_read(h, ms, cnt)
        return {names[i]: (ms[i] / nsteps, cnt[i] // nsteps) for i in range(nk) if cnt[i]}

    # ---------------- stage split (NOT timed): events between the three library calls of a step -------------
    stages = None
    if world == 1 and part is None and args.kktsolver == "chol" and not split_calls and prof:
        nst = 5
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(nst)]
        for q in range(nst):
            bx.blkval.copy_(bx0); by.copy_(by0); L.blkval.copy_(S.blkval)
            ev[q][0].record()
            chordal.cholesky_projected_inverse(L, Y)
            ev[q][1].record()
            solve_ = kkt.factor(L, Y)
            ev[q][2].record()
            solve_(bx, by, 1.0)
            ev[q][3].record()
            if lazy:
                chordal.check_status(symb)
        torch.cuda.synchronize()
        med = lambda a, b: float(np.median([ev[q][a].elapsed_time(ev[q][b]) for q in range(nst)]))
        stages = {"factorisation_ms": round(med(0, 1), 4), "schur_ms": round(med(1, 2), 4), "solve_ms": round(med(2, 3), 4),
                  "what": "median of %d untimed steps, HIP events on the caller's stream between the three library calls of a step: "
                          "csp_cholesky_projected_inverse (scaling point: cholesky + projected_inverse + separator factors), kkt_schur_factor "
                          "(constraint sweeps + Gram accumulation; with the status deferred potrf(H) is left to solve_), kkt_solve (potrf(H) "
                          "beside the first Hessian, two Hessians, Amap / Aadj, potrs)" % nst}
    # Calibration pass (NOT timed): HIP events around every launch give the per-kernel breakdown and name the
    # dominant kernel.  Events around ~100 launches per step cost about 1 ms per step, so the timed region below
    # carries events around the dominant kernel's launches only (csp_profile_filter).
    calib = {}
    if prof:
        lib.csp_profile_filter(h, -1)
        lib.csp_profile_enable(h, 1)
        lib.csp_profile_read(h, None, None)  # clear
        ncal = 2
        for _ in range(ncal):
            step()
        barrier()
        calib = read_profile(ncal)
        dom0 = max(calib, key=lambda k: calib[k][0])
        lib.csp_profile_filter(h, names.index(dom0))
        lib.csp_profile_read(h, None, None)
    if stages is not None and calib:
        stages["gram_kernels_ms"] = round(sum(v[0] for k, v in calib.items() if k.startswith("k_gram") or k.startswith("k_leaf")), 4)
    elapsed = timed_loop(steps)
    ms_per_step = 1e3 * elapsed / steps
    if lazy:
        chordal.lazy_status(symb, False)

    # ---------------- roofline of the dominant kernel (HIP events on its launches inside the timed region) ------
    roofline = None
    breakdown = {}
    if prof:
        timed = read_profile(steps)              # the dominant kernel only, from the timed steps above
        lib.csp_profile_enable(h, 0)
        lib.csp_profile_filter(h, -1)
        breakdown = dict(calib)
        breakdown.update(timed)
        dom = dom0
        dom_ms, dom_launches = breakdown[dom]
        roofline = kernel_roofline(dom, dom_ms, dom_launches, breakdown, symb, m, max_rhs, j1 - j0, part, rank, world, workload,
                                   con=(cptr, cidx))
        if roofline is not None:
            roofline["step"] = step_roofline(workload, ms_per_step, world)
        if args.verbose and rank == 0:
            for k, (t, c) in sorted(breakdown.items(), key=lambda kv: -kv[1][0]):
                print("  %-26s %9.3f ms/step  %5d launches" % (k, t, c), file=sys.stderr)

    # ---------------- back-solve sub-rate: solve_ alone on the factored system of the last step ------
    # two Hessian applications + Amap / Aadj + potrs (solvers.py:506-541); an interior-point iteration issues ~9 of them
    # per factorisation (solvers.py:907-913, 1035-1044)
    if world == 1 and part is None and args.kktsolver == "chol" and not args.no_back_solve:
        nbs = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(nbs + 2):
            if it == 2:
                e0.record()
            bx.blkval.copy_(bx0)
            by.copy_(by0)
            chk(lib.kkt_solve(h, L.blkval.data_ptr(), Y.blkval.data_ptr(), H.data_ptr(), m, 1.0,
                              bx.blkval.data_ptr(), by.data_ptr(), st()), "solve")
        e1.record()
        torch.cuda.synchronize()
        bs_ms = e0.elapsed_time(e1) / nbs
        back_solve = {"ms": round(bs_ms, 4), "per_s": round(1e3 / bs_ms, 2), "calls": nbs,
                      "what": "kkt_solve (solve_, solvers.py:506-541) on the factored system, right-hand sides reset per call"}

    # ---------------- N > 1: the sharded step against the plain single-rank step on rank 0's device (untimed) --------
    # y is replicated, x was left sharded: compared on the cliques rank 0 owns and the top
    sharded_check = None
    if (world > 1 or force_sharded) and part is not None and rank == 0 and primary and args.kktsolver == "chol" and not args.no_check:
        try:
            own = kkt._own_mask.bool().clone()
            for a_, b_ in part.top_ranges:
                own[a_:b_] = True
            xs, ys = bx.blkval.clone(), by.clone()
            single = KKTSystem(symb, cptr, cidx, cval, max_rhs=max_rhs)
            L1 = S.copy()
            chordal.cholesky(L1)
            Y1 = L1.copy()
            chordal.projected_inverse(Y1)
            cx, cy = cspmatrix(symb, bx0.clone()), by0.clone()
            single.factor(L1, Y1)(cx, cy, 1.0)
            mskd = torch.from_numpy(msk).to(dev) & own
            sharded_check = {"x_relerr_on_owned": float("%.2e" % float((xs - cx.blkval).abs()[mskd].max() / cx.blkval.abs()[mskd].max())),
                             "y_relerr": float("%.2e" % float((ys - cy).abs().max() / cy.abs().max())),
                             "what": "search direction of the last timed sharded step against the unsharded step on rank 0's device"}
            del single, L1, Y1, cx, cy
        except Exception as e:      # the check must not take the line down
            sharded_check = {"error": repr(e)}

    # ---------------- CPU baseline: the oracle on a bounded sample (rank 0, N = 1 only) --------
    cpu = None
    # (the oracle leg times the reference's default kkt_chol path and needs the factored H: not run for --kktsolver qr)
    if rank == 0 and world == 1 and want_cpu == "quick":
        from oracle import oracle as orc
        if not orc.use_blas(True):          # plain triple loops on fronts of 64 ... 4096 columns would take minutes: no quick leg
            want_cpu = None
        orc.use_blas(False)
    if rank == 0 and world == 1 and want_cpu and args.kktsolver == "chol":
        from oracle import oracle as orc
        So = orc.Sym(symb)
        K = orc.KKT(So, cptr, cidx, cval)
        Sh = S.blkval.cpu().numpy()
        nthr = max(1, min(args.cpu_threads, os.cpu_count() or 1, m))
        quick = want_cpu == "quick"
        ncols = min(max(args.cpu_cols if not quick else nthr, nthr), m)      # quick: one Schur column per thread, scaled to m
        Hfac = np.asfortranarray(np.tril(H.cpu().numpy().T))   # factored H from the GPU (to time solve_ and potrf)
        Hfull = Hfac @ Hfac.T

        scm = workload == "maxcut"      # every constraint column-sparse: the reference's trsm + SCMcolumn2 route
        if scm:
            nthr = 1
            ncols = min(args.cpu_cols, m)

        def cpu_unit():
            """One Newton-KKT solve on the host: cholesky + projected_inverse (one thread: sequential over the cliques,
            as CHOMPACK is), the Schur columns (one Hessian application per column as in the reference's loop,
            solvers.py:479-487, the independent columns spread over OpenMP threads -- or, for column-sparse constraints,
            two supernodal triangular solves + SCMcolumn2 per column, solvers.py:489-497), potrf(H) and one solve_."""
            t0 = time.perf_counter()
            Lo = Sh.copy()
            orc.cholesky(So, Lo)
            Yo = Lo.copy()
            orc.projected_inverse(So, Yo)
            t_fact = time.perf_counter() - t0
            t0 = time.perf_counter()
            if scm:
                Hc = np.tril(K.schur_scm_columns(Lo, np.zeros((m, m), order="F"), range(ncols)))[:, :ncols]
                t_cols = time.perf_counter() - t0
            elif nthr > 1:
                Hc = K.schur_columns_threaded(Lo, Yo, 0, ncols, nthr)
                t_cols = K.last_seconds            # without the one-off allocation of the per-thread workspaces
            else:
                Hc = K.schur_factor(Lo, Yo, ncols=ncols)[:, :ncols]
                t_cols = time.perf_counter() - t0
            cpu_unit.Hcols = Hc
            Hc = Hfull.copy(order="F")
            t0 = time.perf_counter()
            orc.dense_potrf(Hc)
            t_potrf = time.perf_counter() - t0
            t0 = time.perf_counter()
            xo, yo = K.solve(Lo, Yo, Hfac, bx0.cpu().numpy(), by0.cpu().numpy(), 1.0)
            t_solve = time.perf_counter() - t0
            return t_fact, t_cols, t_potrf, t_solve, xo, yo

        def unit_seconds(t):
            return t[0] + t[1] * (m / ncols) + t[2] + t[3]

        blas_desc = None
        if quick:
            blas_desc = orc.use_blas(True)                              # (the fronts of configs 2 and 3 are 64 ... 4096 columns wide)
        first = cpu_unit()                                             # headline: plain loops, the parity checker
        if quick and blas_desc:
            orc.use_blas(False)
        xo, yo = first[4], first[5]
        t_loops = None if (quick and blas_desc) else unit_seconds(first)
        # the same run doubles as a full-size check of the GPU search direction
        ex = np.linalg.norm((bx.blkval.cpu().numpy() - xo)[msk]) / max(1e-300, np.linalg.norm(xo[msk]))
        ey = np.linalg.norm(by.cpu().numpy() - yo) / max(1e-300, np.linalg.norm(yo))
        # ... and of the Schur complement itself: the oracle's columns against L_H L_H^T of the device's factor (the solve above
        # takes the device's H, so x and y alone would not notice a wrong H)
        Hg = np.tril(Hfull)[:, :ncols] if scm else Hfull[:, :ncols]
        eH = np.linalg.norm(Hg - cpu_unit.Hcols) / max(1e-300, np.linalg.norm(cpu_unit.Hcols))
        reps = [first]
        if not quick:
            blas_desc = orc.use_blas(True)                              # per-clique BLAS-3 on the host BLAS from dimension 32 on
            if blas_desc:
                reps = [cpu_unit() for _ in range(max(1, args.cpu_repeats))]
                orc.use_blas(False)
        reps.sort(key=unit_seconds)
        med = reps[len(reps) // 2]                                      # the median repeat
        t_fact, t_cols, t_potrf, t_solve = med[:4]
        t_unit = unit_seconds(med)
        cpu = {"value": round(1.0 / t_unit, 5), "unit": "KKT solves/s", "cores": nthr, "threads": nthr, "host_cores": os.cpu_count(),
               "kind": "port", "repeats": len(reps), "value_min_max": [round(1.0 / unit_seconds(reps[-1]), 5), round(1.0 / unit_seconds(reps[0]), 5)],
               "blas": blas_desc or "none (plain loops)", "value_plain_loops": round(1.0 / t_loops, 5) if t_loops else None,
               "sample": "median of %d repeats of: cholesky+projected_inverse (%.3fs, 1 thread: sequential over the cliques as CHOMPACK is) + "
                         "%d of %d Schur columns on %d thread(s) of the host's %d cores (%.3fs%s%s) + potrf(H) (%.4fs) + 1 solve_ (%.3fs, 1 thread); "
                         "oracle/chordal_oracle.c with per-clique dense operations of dimension >= 32 on %s"
                         % (len(reps), t_fact, ncols, m, nthr, os.cpu_count(), t_cols, "" if ncols == m else ", scaled x%.2f" % (m / ncols),
                            "; trsm x 2 + SCMcolumn2 per column, solvers.py:489-497" if scm else "", t_potrf,
                            t_solve, blas_desc or "plain loops")}
        cpu["gpu_vs_oracle_relerr"] = [float("%.2e" % ex), float("%.2e" % ey)]
        cpu["schur_vs_oracle_relerr"] = float("%.2e" % eH)
        if scm:
            cpu["cores_note"] = ("one host thread: the reference's SCMcolumn2 route (solvers.py:489-497) is a sequential Python loop over "
                                 "the columns, two supernodal triangular solves + one misc.SCMcolumn2 call each on n = 1000 -- BLAS-2 sized "
                                 "work that a threaded BLAS does not speed up; the oracle restates that loop as it is")

    result = None
    if rank == 0:
        out = {
            "metric": "Newton KKT solves/sec", "value": round(steps / elapsed, 4), "unit": "KKT solves/s",
            "value_tuned": round(steps / elapsed, 4) if value_as_allocated is not None else None,
            "value_as_allocated": round(value_as_allocated, 4) if value_as_allocated is not None else round(steps / elapsed, 4),
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": label, "kktsolver": args.kktsolver, "n": symb.n, "cliques": symb.Nsn, "m": m, "blkval_doubles": int(B),
                       "update_doubles": int(U), "rhs_per_sweep": max_rhs, "placement_tuning": placement,
                       "parallelism": ("single" if world == 1 else
                                       ("subtree-sharded Gram + boundary exchange/%d" % world if part is not None
                                        else ("column-sparse constraints by rank (SCMcolumn2)/%d" % world if workload == "maxcut"
                                              else "schur-columns/%d" % world)))},
            "roofline": roofline, "stages": stages, "cpu_baseline": cpu, "back_solve": back_solve, "sharded_vs_single": sharded_check,
            "symbolic_s": round(t_sym, 3), "csrc_sha256": csrc_sha256(),
            # per-kernel HIP-event times: the dominant kernel from the timed steps, the others from the untimed
            # calibration pass that precedes them (events around every launch)
            "kernel_ms_per_step": {k: round(v[0], 4) for k, v in sorted(breakdown.items(), key=lambda kv: -kv[1][0])},
        }
        result = out
    # release the device context of this workload before the next one is built
    if lazy:
        pass
    del kkt, L, Y, S, bx, by, H
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return result


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="synth50k")
    ap.add_argument("--kktsolver", default="chol", choices=("chol", "qr"),
                    help="'qr': the step factors with kkt_qr (Cholesky-QR of the swept stack) instead of kkt_chol; one GPU")
    ap.add_argument("--m", type=int, default=None)
    ap.add_argument("--max-rhs", type=int, default=None)
    ap.add_argument("--cpu-cols", type=int, default=10 ** 9, help="Schur columns timed on the CPU oracle (default: all; at least one per thread)")
    ap.add_argument("--cpu-threads", type=int, default=16, help="host threads of the CPU baseline (a 1-GPU box has a 16-core share)")
    ap.add_argument("--tune-placement", type=int, default=0,
                    help="tries of csp_tune(CSP_TUNE_PLACEMENT) during set-up (0 = off, the default since round 4: with the fused extend-add the "
                         "family sweep no longer writes the packed exchange buffer, and the placement of its buffers stopped mattering -- 301 solves/s "
                         "as allocated against 298 tuned; synth50k only): the two output buffers of the family sweep are moved in turn to fresh "
                         "allocations, the fastest for its store pattern kept; `value_as_allocated` and `value_tuned` are then both reported")
    ap.add_argument("--shard", default="subtree", choices=["subtree", "columns"],
                    help="N > 1: subtree sharding + boundary exchange (default) or column sharding of H")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="N > 1: skip the (untimed) comparison of the sharded step with the single-rank step")
    ap.add_argument("--no-back-solve", action="store_true", help="skip the solve_-only timing loop (kernel traces of exactly the timed steps)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the GPU-only figures of configs 2, 3 and 4 after the headline run")
    ap.add_argument("--cpu-repeats", type=int, default=3, help="repeats of the CPU baseline (the median is reported)")
    ap.add_argument("--no-profile", action="store_true", help="skip HIP-event kernel timing")
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()

    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        # Not under a launcher: start the N ranks as fresh child processes (one per GPU, RCCL) BEFORE this process
        # touches the GPU, relay rank 0's JSON line and exit with the launcher's code.  Nothing is exec'd.
        raise SystemExit(self_launch(args.gpus))
    if int(world_env or "1") != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%s; launch with `python bench.py --gpus N` or "
              "`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`" % (args.gpus, world_env),
              file=sys.stderr)
        raise SystemExit(2)

    # RCCL prints a version banner on stdout when its first communicator comes up: everything but the result line goes
    # to stderr (file descriptor 1 points at stderr from here on; rank 0 writes the JSON line to the saved stdout)
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from smcp_amd import _lib, chordal, problems
    from smcp_amd.cspmatrix import cspmatrix
    from smcp_amd.kkt import KKTSystem
    from smcp_amd.symbolic import Symbolic

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("SMCP_BENCH_BACKEND", "nccl")   # "gloo": functional check with ranks sharing a GPU
        ndev = torch.cuda.device_count()
        torch.cuda.set_device(local % ndev)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local % ndev))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    # SMCP_BENCH_FORCE_SHARDED=1 (N = 1 only): the subtree-sharded route and its collectives over RCCL with a group of ONE
    # rank -- what the host logic and the collective launches of the N-GPU step cost beside the plain single-GPU step
    force_sharded = world == 1 and os.environ.get("SMCP_BENCH_FORCE_SHARDED") == "1"
    if force_sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    dev = torch.device("cuda", torch.cuda.current_device())
    lib = _lib.lib()

    env = {"world": world, "rank": rank, "dev": dev, "lib": lib, "force_sharded": force_sharded}
    out = run_workload(args, args.workload, args.steps, args.warmup, None if args.no_cpu else "full", True, env)
    # BASELINE.json's other single-GPU configurations, GPU only: config 2 (one dense 4096 front), config 3 (block-arrow),
    # config 4 (max-cut, column-sparse constraints) -- a few steps each after the headline measurement
    if world == 1 and not force_sharded and args.workload == "synth50k" and not args.no_secondary and args.kktsolver == "chol":
        sec = {}
        for name in ("dense4096", "arrow", "maxcut", "synth50k_dense", "synth50k_trace", "band200"):
            try:
                # ten timed steps after two warm-up steps each; the CPU leg beside every GPU figure: the oracle on the host BLAS, one
                # Schur column per thread scaled to m (config 4: its sequential SCMcolumn2 route in full, median of three)
                cpu_mode = None if args.no_cpu else ("full" if name == "maxcut" else "quick")
                r = run_workload(args, name, 10, 2, cpu_mode, False, env)
                sec[name] = {k: r[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "config", "roofline", "stages", "back_solve", "cpu_baseline")}
                sec[name]["top_kernels_ms_per_step"] = dict(list(r["kernel_ms_per_step"].items())[:6])
            except Exception as e:      # a secondary workload must not take the headline line down
                sec[name] = {"error": repr(e)}
        if out is not None:
            out["secondary"] = sec
    if rank == 0 and out is not None:
        print(json.dumps(out), file=real_stdout, flush=True)
    if world > 1:
        dist.barrier()          # rank 0 may still be in its untimed checks
    if world > 1 or force_sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

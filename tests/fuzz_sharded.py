"""Sharded factorisation + Schur complement + solve_ on random clique trees, `world` ranks sharing ONE GPU over gloo, against
the single-rank results of the same device code (tests/test_gpu_distributed.py; long runs: python scratch/fuzz_sharded.py
[ncases] [seed0] [world])."""
import os, socket, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port, ncases, seed0, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        import fuzz_parity
        from smcp_amd import chordal, problems
        from smcp_amd.cspmatrix import cspmatrix
        from smcp_amd.kkt import KKTSystem
        from smcp_amd.symbolic import Symbolic
        bad = 0
        only = os.environ.get("SMCP_FUZZ_ONLY")            # one case of a long run on its own (same pattern kind, same seed)
        for case in range(ncases):
            if only is not None and case != int(only):
                continue
            rng = np.random.default_rng(seed0 + case)
            symb = Symbolic(fuzz_parity.pattern(rng, case))
            m = int(rng.integers(2, 14))
            mr = int(rng.integers(2, 6))
            symb.device_init(0, mr)
            S = cspmatrix(symb, torch.from_numpy(problems.random_factor_blkval(symb, int(rng.integers(1 << 30)))).cuda())
            chordal.llt(S)
            msk = np.zeros(symb.blklen, dtype=bool); msk[problems.lower_positions(symb)] = True
            nnzv = int(msk.sum()); m = min(m, max(1, nnzv // 3))
            cptr, cidx, cval = problems.random_constraints(symb, m, density=float(rng.choice([0.01, 0.05, 0.3])), seed=int(rng.integers(1 << 30)))
            b0 = torch.from_numpy(rng.standard_normal(symb.blklen) * msk).cuda()
            y0 = torch.from_numpy(rng.standard_normal(m)).cuda()
            tag = "case %d kind %d n=%d nsn=%d maxnn=%d maxna=%d m=%d rhs=%d" % (seed0 + case, case % 6, symb.n, symb.Nsn, symb.max_nn, symb.max_na, m, mr)
            if rank == 0 and os.environ.get("SMCP_FUZZ_TRACE"):
                print(tag, flush=True)
            # single-rank reference on this rank's own context (tnzcols = 0: the sharded paths sweep every constraint)
            L1 = S.copy(); chordal.cholesky(L1); Y1 = L1.copy(); chordal.projected_inverse(Y1)
            single = KKTSystem(symb, cptr, cidx, cval, max_rhs=mr, tnzcols=0.0)
            try:
                solve1 = single.factor(L1, Y1)
            except ArithmeticError:
                continue                                   # dependent constraints: not a case
            H1 = single.H.clone()
            cx, cy = cspmatrix(symb, b0.clone()), y0.clone()
            solve1(cx, cy, 0.6)
            sh = KKTSystem(symb, cptr, cidx, cval, max_rhs=mr, tnzcols=0.0)
            chordal.tune(symb, chordal.TUNE_LEAFGRAM, 2 if case % 2 == 0 else 1)     # closed-form leaf Gram blocks, sharded too
            P = sh.set_partition(dist.group.WORLD)
            if os.environ.get("SMCP_FS_MODE") == "gram":        # replicated factor, sharded Schur complement only
                L, Y = L1, Y1
            else:
                L, Y = sh.factor_scaling(S, dist.group.WORLD)
            own = sh._own_mask.bool().clone()
            for a, b in P.top_ranges:
                own[a:b] = True
            mskd = torch.from_numpy(msk).cuda()
            rel = lambda a, b, w: float((a - b).abs()[w].max() / max(float(b.abs().max()), 1e-300)) if bool(w.any()) else 0.0
            solve = sh.factor(L, Y, group=dist.group.WORLD)
            bx, by = cspmatrix(symb, b0.clone()), y0.clone()
            solve(bx, by, 0.6)
            errs = dict(L=rel(L.blkval, L1.blkval, own & mskd), Y=rel(Y.blkval, Y1.blkval, own & mskd),
                        H=float((sh.H - H1).abs().max() / H1.abs().max()), x=rel(bx.blkval, cx.blkval, mskd),
                        y=float((by - cy).abs().max() / cy.abs().max()))
            # H1 holds the Cholesky factor of the Schur complement after factor(): cond(H) = cond(factor)^2.  (The symmetrised
            # factor was priced here until round 3: nearly dependent constraints whose potrf barely goes through -- y ~ 1e17
            # on both sides -- then look like a mismatch of x; seed 93098.)
            cond = float(torch.linalg.cond(torch.tril(H1))) ** 2 if m > 1 else 1.0
            if not cond < 1e13:
                continue                                   # numerically dependent constraints: not a case
            tol = 1e-10 * max(1.0, cond / 1e4)
            worst = max(errs.values())
            if not worst < tol:
                bad += 1
                if rank == 0:
                    print("MISMATCH", tag, {k: "%.1e" % v for k, v in errs.items()}, "cond %.1e" % cond, "top", len(P.top), flush=True)
            elif rank == 0 and case % 10 == 9:
                print("ok through", tag, flush=True)
        if rank == 0:
            out.put(bad)
    finally:
        dist.destroy_process_group()


def main(ncases=30, seed0=5000, world=2):
    """Returns the number of mismatching cases (-1: a rank died)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    procs = [ctx.Process(target=worker, args=(r, world, port, ncases, seed0, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join()
    codes = [p.exitcode for p in procs]
    return out.get() if all(c == 0 for c in codes) else -1


if __name__ == "__main__":
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    world = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    bad = main(ncases, seed0, world)
    print("cases %d mismatches %d" % (ncases, bad))
    sys.exit(0 if bad == 0 else 1)

"""Debug of fuzz_sharded case 98 of seed0 93000 (two gloo ranks on one GPU): where does x differ?"""
import os, socket, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch, torch.distributed as dist, torch.multiprocessing as mp

def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import fuzz_parity
    from smcp_amd import chordal, problems
    from smcp_amd.cspmatrix import cspmatrix
    from smcp_amd.kkt import KKTSystem
    from smcp_amd.symbolic import Symbolic
    case, seed0 = 98, 93000
    rng = np.random.default_rng(seed0 + case)
    symb = Symbolic(fuzz_parity.pattern(rng, case))
    m = int(rng.integers(2, 14)); mr = int(rng.integers(2, 6))
    symb.device_init(0, mr)
    S = cspmatrix(symb, torch.from_numpy(problems.random_factor_blkval(symb, int(rng.integers(1 << 30)))).cuda())
    chordal.llt(S)
    msk = np.zeros(symb.blklen, dtype=bool); msk[problems.lower_positions(symb)] = True
    nnzv = int(msk.sum()); m = min(m, max(1, nnzv // 3))
    cptr, cidx, cval = problems.random_constraints(symb, m, density=float(rng.choice([0.01, 0.05, 0.3])), seed=int(rng.integers(1 << 30)))
    b0 = torch.from_numpy(rng.standard_normal(symb.blklen) * msk).cuda()
    y0 = torch.from_numpy(rng.standard_normal(m)).cuda()
    L1 = S.copy(); chordal.cholesky(L1); Y1 = L1.copy(); chordal.projected_inverse(Y1)
    single = KKTSystem(symb, cptr, cidx, cval, max_rhs=mr, tnzcols=0.0)
    solve1 = single.factor(L1, Y1)
    cx, cy = cspmatrix(symb, b0.clone()), y0.clone()
    solve1(cx, cy, 0.6)
    # plain Hessian of b0 on the single-rank path, for the first-Hessian comparison
    h1 = cspmatrix(symb, b0.clone()); chordal.hessian(L1, Y1, h1, adj=None, inv=False)
    sh = KKTSystem(symb, cptr, cidx, cval, max_rhs=mr, tnzcols=0.0)
    chordal.tune(symb, chordal.TUNE_LEAFGRAM, 2)
    P = sh.set_partition(dist.group.WORLD)
    L, Y = sh.factor_scaling(S, dist.group.WORLD)
    solve = sh.factor(L, Y, group=dist.group.WORLD)
    nn, na = symb.clique_sizes()
    if rank == 0:
        print("cliques (nn, na):", list(zip(nn.tolist(), na.tolist())), "owner", P.owner.tolist(), "fam", symb.family_roles().tolist() if hasattr(symb, "family_roles") else None, flush=True)
    # first Hessian alone through the sharded halves
    U = cspmatrix(symb, b0.clone())
    sh._hess_part(U, 1, 0); sh._exchange(dist.group.WORLD, 1); sh._hess_part(U, 2, 0); sh._hess_part(U, 2, 1); sh._hess_part(U, 1, 1)
    own = sh._own_mask.bool().clone()
    for a, b in P.top_ranges: own[a:b] = True
    mskd = torch.from_numpy(msk).cuda()
    d = (U.blkval - h1.blkval).abs() * (own & mskd)
    print("rank", rank, "first Hessian on owned+top: max err %.2e at %d (scale %.2e)" % (float(d.max()), int(d.argmax()), float(h1.blkval.abs().max())), flush=True)
    bx, by = cspmatrix(symb, b0.clone()), y0.clone()
    solve(bx, by, 0.6)
    if rank == 0:
        Hs = torch.tril(single.H) + torch.tril(single.H, -1).T
        print("cy", cy.cpu().numpy(), "\nby", by.cpu().numpy(), "\neig H1", torch.linalg.eigvalsh(Hs).cpu().numpy(), "\ny0", y0.cpu().numpy(), flush=True)
    d = (bx.blkval - cx.blkval).abs() * mskd
    blk = np.asarray(symb._query("blkptr") if hasattr(symb, "_query") else [])
    print("rank", rank, "x: max err %.2e at %d (scale %.2e); y err %.2e" % (float(d.max()), int(d.argmax()), float(cx.blkval.abs().max()), float((by - cy).abs().max())), flush=True)
    for k in range(symb.Nsn):
        pass
    dist.destroy_process_group()

if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=worker, args=(r, 2, port)) for r in range(2)]
    for p in procs: p.start()
    for p in procs: p.join()

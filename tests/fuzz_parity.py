"""Randomised parity sweep (GPU vs oracle): random clique trees of many shapes (random trees, nested block-arrow,
band, block-arrow, fat cliques beyond LDS), with and without amalgamation; every tree operation, every Hessian mode
and the KKT solvers (kkt_chol with three constraint classifications, kkt_qr) against the oracle.
Used by tests/test_gpu_fuzz.py (a few cases) and scratch/fuzz_parity.py (long sweeps)."""
import os
import time

import numpy as np
import torch
from oracle import oracle as orc
from smcp_amd import problems, chordal
from smcp_amd.symbolic import Symbolic, amalgamate
from smcp_amd.cspmatrix import cspmatrix
from smcp_amd.kkt import KKTSystem

rel = lambda a, b: np.linalg.norm(a - b) / max(1e-300, np.linalg.norm(b))

def pattern(rng, case):
    kind = case % 6
    if kind == 0:
        return problems.random_chordal_pattern(int(rng.integers(5, 120)), max_nn=int(rng.integers(1, 20)),
                                               max_na=int(rng.integers(1, 40)), seed=int(rng.integers(1 << 30)))
    if kind == 1:
        return nested(rng, (1, 17), (1, 33), (1, 17), (1, 65))
    if kind == 2:
        return problems.band_pattern(int(rng.integers(8, 150)), int(rng.integers(0, 7)))
    if kind == 3:
        return problems.block_arrow_pattern(int(rng.integers(1, 12)), int(rng.integers(1, 70)), int(rng.integers(0, 140)))
    if kind == 4:
        return problems.random_chordal_pattern(int(rng.integers(3, 30)), max_nn=int(rng.integers(20, 80)),
                                               max_na=int(rng.integers(30, 200)), seed=int(rng.integers(1 << 30)))
    return nested(rng, (1, 17), (1, 33), (17, 70), (33, 130))

def nested(rng, ln, la, mn, ma):
    root = int(rng.integers(8, 150))
    tn, ta = int(rng.integers(8, 70)), int(rng.integers(1, root + 1))
    mn_, ma_ = int(rng.integers(*mn)), int(rng.integers(ma[0], ma[1]))
    ma_ = max(1, min(ma_, tn + ta))
    ln_, la_ = int(rng.integers(*ln)), int(rng.integers(*la))
    la_ = max(1, min(la_, mn_ + ma_))
    return problems.nested_block_arrow_pattern(nsub=int(rng.integers(1, 3)), nmid=int(rng.integers(1, 6)),
                                               nleaf_per_mid=int(rng.integers(1, 11)), leaf=(ln_, la_), mid=(mn_, ma_),
                                               top=(tn, ta), root=root, seed=int(rng.integers(1 << 30)))

def run(ncases, seed0=0, verbose=False, patterns=None):
    """Returns {check: (worst relative error, case description)} over ncases random problems (or over the given
    patterns, one case each)."""
    worst = {}

    def note(k, v, tag):
        if not (v <= worst.get(k, (-1.0, ""))[0]):
            worst[k] = (float(v), tag)

    t0 = time.time()
    for case in range(ncases):
        rng = np.random.default_rng(seed0 + case)
        pat = patterns[case] if patterns is not None else pattern(rng, case)
        symb = Symbolic(pat)
        if patterns is None and rng.random() < 0.4:
            emb = amalgamate(symb)
            if emb is not None:
                symb = Symbolic(emb[0], emb[1])
        tag = "case %d kind %d n=%d nsn=%d maxnn=%d maxna=%d" % (seed0 + case, case % 6, symb.n, symb.Nsn, symb.max_nn, symb.max_na)
        if os.environ.get("SMCP_FUZZ_TRACE"):        # the case about to run (a device fault leaves no other trace)
            print(tag, flush=True)
        m = int(rng.integers(1, 20))
        nrhs = int(rng.integers(1, 6))
        symb.device_init(0, max(nrhs, min(m, int(rng.integers(1, 8)))))
        S = orc.Sym(symb)
        msk = np.zeros(symb.blklen, dtype=bool); msk[problems.lower_positions(symb)] = True
        dev = lambda x: cspmatrix(symb, torch.from_numpy(np.ascontiguousarray(x)).cuda())
        host = lambda X: X.blkval.cpu().numpy()
        Lh = problems.random_factor_blkval(symb, int(rng.integers(1 << 30)))
        A = Lh.copy(); orc.llt(S, A)
        # cholesky / projected inverse / completion
        X = dev(A); chordal.cholesky(X)
        Lr = A.copy(); orc.cholesky(S, Lr)
        note("cholesky", rel(host(X)[msk], Lr[msk]), tag)
        Y = X.copy(); chordal.projected_inverse(Y)
        Yr = Lr.copy(); orc.projected_inverse(S, Yr)
        note("projected_inverse", rel(host(Y)[msk], Yr[msk]), tag)
        C = Y.copy(); chordal.completion(C)
        Cr = Yr.copy(); orc.completion(S, Cr)
        note("completion", rel(host(C)[msk], Cr[msk]), tag)
        # hessian, all modes
        U = rng.standard_normal((nrhs, symb.blklen)) * msk
        for adj in (None, False, True):
            for inv in (False, True):
                ref = U.copy()
                for r in range(nrhs):
                    orc.hessian(S, Lr, Yr, ref[r], adj=adj, inv=inv)
                Ud = torch.from_numpy(U.copy()).cuda()
                chordal.hessian(dev(Lr), dev(Yr), Ud, adj=adj, inv=inv)
                got = Ud.cpu().numpy()
                note("hessian adj=%s inv=%s" % (adj, inv), max(rel(got[r][msk], ref[r][msk]) for r in range(nrhs)), tag)
        # KKT: chol (with and without the column-sparse split) and qr
        nnzv = int(msk.sum())
        m = min(m, max(1, nnzv // 2))
        dens = float(rng.choice([0.002, 0.02, 0.2]))
        cptr, cidx, cval = problems.random_constraints(symb, m, density=dens, seed=int(rng.integers(1 << 30)))
        if case % 3 == 0 and m >= 2:
            # a long list among short ones (round 5): one or two constraints become a band of V -- width 0 is a diagonal matrix,
            # the trace constraint of a relaxation -- whose (family, constraint) lists take several chunks of the entry-driven sweeps
            ccp, cri = symb.sparsity_pattern()
            ccol = np.repeat(np.arange(symb.n), np.diff(ccp))
            cols = [(cidx[cptr[j]:cptr[j + 1]], cval[cptr[j]:cptr[j + 1]]) for j in range(m)]
            for j in rng.choice(m, size=int(rng.integers(1, 3)), replace=False):
                pos = np.sort(symb.ccs_to_blk()[(cri - ccol) <= int(rng.integers(0, 5))]).astype(np.int64)
                cols[int(j)] = (pos, rng.standard_normal(len(pos)))
            cptr = np.concatenate([[0], np.cumsum([len(p_) for p_, _ in cols])]).astype(np.int64)
            cidx = np.concatenate([p_ for p_, _ in cols]).astype(np.int64)
            cval = np.concatenate([v_ for _, v_ in cols])
            tag = tag + " +band" if isinstance(tag, str) else tag
        K = orc.KKT(S, cptr, cidx, cval)
        try:
            Href = K.schur_factor(Lr, Yr)
        except Exception:
            continue            # dependent constraints (tiny patterns): not a parity case
        bx = rng.standard_normal(symb.blklen) * msk
        by = rng.standard_normal(m)
        kk = float(rng.choice([1.0, 0.3, 7.0]))
        xr, yr = K.solve(Lr, Yr, Href, bx, by, kk)
        if not np.isfinite(xr).all() or np.linalg.cond(np.tril(Href)) > 1e6:
            continue
        # closed-form Gram blocks of family children + the entry-driven family sweep (round 3): forced on every other case
        # (the cost rule would rarely pick them on trees this small), off otherwise
        chordal.tune(symb, chordal.TUNE_LEAFGRAM, 2 if case % 2 == 0 else 1)
        for tnz in (None, 0.0, 1.0):
            sysk = KKTSystem(symb, cptr, cidx, cval, max_rhs=symb._max_rhs, tnzcols=tnz)
            solve = sysk.factor(dev(Lr), dev(Yr))
            bxd, byd = dev(bx), torch.from_numpy(by.copy()).cuda()
            solve(bxd, byd, kk)
            note("kkt_chol tnz=%s" % tnz, max(rel(host(bxd)[msk], xr[msk]), rel(byd.cpu().numpy(), yr)), tag)
        sysk = KKTSystem(symb, cptr, cidx, cval, max_rhs=symb._max_rhs, tnzcols=0.0)
        solve = sysk.factor_qr(dev(Lr), dev(Yr))
        bxd, byd = dev(bx), torch.from_numpy(by.copy()).cuda()
        solve(bxd, byd, kk)
        note("kkt_qr", max(rel(host(bxd)[msk], xr[msk]), rel(byd.cpu().numpy(), yr)), tag)
        if verbose and case % 10 == 9:
            print("case", case, "%.0f s" % (time.time() - t0), flush=True)
    return worst

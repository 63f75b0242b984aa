"""Interior-point drivers for sparse matrix cone programs on top of the HIP chordal kernels.

Host-side (Python) counterparts of the reference's drivers; the loop stays in Python and every
chordal operation goes through the C-ABI (include/smcp_amd.h):

  chordalsolver_esd   extended self-dual embedding      (reference: src/python/solvers.py:1330-2467)
  chordalsolver_feas  feasible-start barrier method      (reference: src/python/solvers.py:49-1327)
  conelp / lp / socp / sdp   CVXOPT-style front ends     (reference: src/python/solvers.py:2470-2699)

Problem format (solvers.py:54-62 docstring): ``A`` is an n^2 x (m+1) sparse matrix whose columns
are vec(C), vec(A_1), ..., vec(A_m), lower triangles only; ``b`` has length m.
cvxopt is not a dependency here: sparse inputs/outputs are scipy.sparse, dense ones numpy.
"""
import copy as _copy
import math
import time

import numpy as np
import scipy.sparse as sp
import torch

from . import chordal
from .cspmatrix import cspmatrix, has_device
from .kkt import KKTSystem
from .symbolic import Symbolic, amalgamate, maxcardsearch, mindegree

# same keys / defaults as the reference (solvers.py:22-44)
options = {
    "debug": False, "maxiters": 100, "abstol": 1e-6, "reltol": 1e-6, "feastol": 1e-8, "refinement": 2,
    "cholmod": False, "order": "AMD", "tnzcols": 0.1, "show_progress": True, "dimacs": True, "eta": None,
    "delta": 0.9, "alpha": 1e-1, "beta": 0.7, "minstep": 1e-8, "lifting": True, "t0": 1e-1,
    "equalsteps": True, "prediction": True, "step": 0.98,
    # Two additions for chordalsolver_esd (not in the reference).  The reference refines only the outer
    # 5-block Newton system and forms dS through the inverse Hessian (solvers.py:2017-2022, 2082-2086); with
    # exactly that scheme the normal-equations solve loses A*x = by to cancellation once t ~ 1/mu is large
    # (x = t*H(A'y - bx)) and the iterates stall at feasibility residuals of 1e-6..1e-7 on the band problems
    # (same behaviour with the CPU oracle as backend).  One refinement step on the 2x2 KKT solve and dS from
    # the dual-feasibility row (the variant the reference keeps commented out at solvers.py:2014-2016)
    # restore convergence to the default tolerances in ~20 iterations.  Set (0, True) for the reference's scheme.
    "esd_kkt_refinement": 1, "esd_ds_from_hessian": False,
    # which perfect elimination order a chordal pattern is analysed in (not in the reference, not observable in the results):
    # 'auto' = the order the pattern is given in when that has zero fill, 'mcs' = maximum cardinality search as the reference
    "peo": "auto",
    # relaxed supernode amalgamation of deep, thin clique trees (smcp_amd.symbolic.amalgamate; not in the reference)
    "amalgamate": True,
    # device-resident line search: the bisection probes of the feasible-start solver as concurrent trial factorisations
    "batched_linesearch": True,
    # ... as long as one trial leaves room on the device: patterns with more than this many stored entries take the trials of
    # each bisection round one after the other (three instead of eight; same trial points, same result)
    "batched_linesearch_maxlen": 1 << 20,
}
_defaults = _copy.deepcopy(options)


def _opt(name, kind, lo=None, hi=None, strict_lo=False):
    v = options.get(name, _defaults[name])
    if kind is bool:
        if not isinstance(v, bool):
            raise TypeError("options['%s'] must be a bool" % name)
        return v
    if kind is int:
        if not isinstance(v, (int, np.integer)) or isinstance(v, bool):
            raise TypeError("options['%s'] must be an integer" % name)
    else:
        if not isinstance(v, (int, float, np.floating, np.integer)) or isinstance(v, bool):
            raise TypeError("options['%s'] must be a scalar" % name)
    if lo is not None and (v < lo or (strict_lo and v <= lo)):
        raise ValueError("options['%s'] out of range" % name)
    if hi is not None and v > hi:
        raise ValueError("options['%s'] out of range" % name)
    return v


class _Problem:
    """Index algebra of solvers.py:234-367 on the C-ABI symbolic layer: aggregate sparsity,
    ordering (perfect elimination order if chordal, else a fill-reducing embedding), C and the
    constraints in blkval coordinates, Amap / Aadj on the device."""

    def __init__(self, A, b, p=None, device=None):
        A = sp.csc_matrix(A)
        self.m = A.shape[1] - 1
        self.n = int(round(math.sqrt(A.shape[0])))
        if self.n * self.n != A.shape[0]:
            raise ValueError("A must have n^2 rows")
        n, m = self.n, self.m
        b = np.asarray(b, dtype=np.float64).reshape(-1)
        if b.shape[0] != m:
            raise ValueError("b must have length m")
        A.sum_duplicates()
        rows = A.indices.astype(np.int64)
        J, I = np.divmod(rows, n)                      # row index of A = i + n j
        if (I < J).any():
            raise ValueError("only lower-triangular entries (i >= j) are allowed in A")
        # aggregate sparsity pattern + diagonal (the key j n + i of an entry IS its row index in A)
        key = np.unique(np.concatenate([rows, np.arange(n, dtype=np.int64) * (n + 1)]))
        pj, pi = np.divmod(key, n)
        cp = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(np.bincount(pj, minlength=n), out=cp[1:])
        pat = (n, cp, pi)
        if p is None:
            # Which perfect elimination order is used is not observable (results are un-permuted, solvers.py:1297-1305), but the
            # clique tree it induces is what the device sweeps level by level: a pattern that is chordal in the order it is GIVEN
            # (generators and modelling tools usually emit their blocks that way) keeps its natural nesting -- on the n = 50 000
            # benchmark pattern four levels where maximum cardinality search leaves seven.  options['peo']: 'auto' (natural
            # order when it has zero fill, else as the reference: maximum cardinality search, then minimum degree,
            # solvers.py:301-308), 'mcs' (the reference's sequence only)
            symb = None
            if options.get("peo", "auto") == "auto":
                symb = Symbolic(pat, None)
                if symb.fill > 0:
                    symb = None
            if symb is None:
                p = maxcardsearch(pat)
                symb = Symbolic(pat, p)
                if symb.fill > 0:                      # not chordal: embed (solvers.py:278-279, 305-308)
                    p = mindegree(pat)
                    symb = Symbolic(pat, p)
        else:
            symb = Symbolic(pat, np.asarray(p, dtype=np.int64))
        self.ischordal = symb.fill == 0
        # deep, thin clique trees are launch-bound: merge runs of small cliques (relaxed supernodes, a chordal
        # embedding with a few explicit zeros; off with options['amalgamate'] = False)
        self.amalgamated = False
        if options.get("amalgamate", True):
            emb = amalgamate(symb)
            if emb is not None:
                symb = Symbolic(emb[0], emb[1])
                self.amalgamated = True
        self.symb = symb
        if torch.cuda.is_available():
            self.dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        else:
            self.dev = torch.device("cpu")   # storage only: every kernel call fails without a GPU
        if m > symb.nnz:                              # solvers.py:351-352
            raise ValueError("more constraints than nonzeros")
        colptr = A.indptr.astype(np.int64)
        pos = symb.index_map(I, J)
        vals = A.data.astype(np.float64)
        # C
        sl = slice(colptr[0], colptr[1])
        h = np.bincount(pos[sl], weights=vals[sl], minlength=symb.blklen).astype(np.float64, copy=False)
        cptr = colptr[1:] - colptr[1]
        tnz = options.get("tnzcols", 0.1)
        if type(tnz) is not float:
            raise TypeError("tnzcols must be a float between 0.0 and 1.0")
        if tnz > 1 or tnz < 0:
            raise ValueError("tnzcols must be between 0.0 and 1.0")
        self._con = (cptr, pos[colptr[1]:], vals[colptr[1]:])
        self._tnz_user = tnz
        self._tnz = tnz
        self.kktsolver = "chol"
        self.kkt = KKTSystem(symb, *self._con, tnzcols=tnz)
        self.C = cspmatrix(symb, torch.from_numpy(h).to(self.dev))
        self.b = torch.from_numpy(b.copy()).to(self.dev)
        self.bh = b
        self.cmaxabs = float(np.abs(vals[sl]).max()) if sl.stop > sl.start else 0.0
        self._diag_idx = None

    def identity(self):
        s = self.symb
        X = cspmatrix(s, torch.zeros(s.blklen, dtype=torch.float64, device=self.dev))
        if self._diag_idx is None:
            nn, na = s.clique_sizes()
            nf = nn + na
            idx = np.concatenate([s.blkptr[k] + np.arange(nn[k]) * (nf[k] + 1) for k in range(s.Nsn)])
            self._diag_idx = torch.as_tensor(idx, device=self.dev)
        X.blkval[self._diag_idx] = 1.0
        return X

    def from_sym(self, M):
        """cspmatrix of a symmetric scipy/numpy matrix given in ORIGINAL coordinates (start points)."""
        M = sp.tril(sp.coo_matrix(M)).tocoo()
        return cspmatrix.from_entries(self.symb, M.row, M.col, M.data, device=self.dev)

    def use_kktsolver(self, name):
        """'chol' (kkt_chol, solvers.py:477-541) or 'qr' (kkt_qr, solvers.py:413-475).  The QR path sweeps every
        constraint -- the reference only splits off column-sparse constraints for 'chol' (solvers.py:242, 355) --
        so the constraint classification is redone with tnzcols = 0 when the solver changes."""
        if name not in ("chol", "qr"):
            raise ValueError("Unknown 'kktsolver'.")
        want = 0.0 if name == "qr" else self._tnz_user
        if want != self._tnz:
            self.kkt = KKTSystem(self.symb, *self._con, tnzcols=want)
            self._tnz = want
        self.kktsolver = name

    def factor(self, L, Y):
        return self.kkt.factor_qr(L, Y) if self.kktsolver == "qr" else self.kkt.factor(L, Y)

    def Amap(self, X):
        return self.kkt.amap(X)

    def Aadj(self, y):
        return self.kkt.aadj(y)

    def to_scipy(self, X):
        """Symmetric scipy matrix in original coordinates (perm(symmetrize(X), ip), solvers.py:2437)."""
        return X.spmatrix(reordered=False, symmetric=True)


def _dot(a, b):
    return float(torch.dot(a, b).item())


def _nrm2(a):
    return float(torch.linalg.vector_norm(a).item())


def _batched():
    return bool(options.get("batched_linesearch", True)) and (has_device()
                                                               or getattr(chordal, "_probe_emulated", False))


def _backtrack(base, d, which, s, beta, smin, accept=None, batch=8, strict=False):
    """The reference's backtracking loops (solvers.py:928-939, 2172-2209): the first step of s, s*beta, s*beta^2, ...
    (while >= smin) for which base + step * d is inside the cone `which` ('p': completion, 'd': cholesky) and
    accept(factor, step) holds (if given).  Returns the step, or None when the steps run out.
    The trial points are the reference's and are examined in the reference's order, so the result is the same;
    with options['batched_linesearch'] everything after a first failure is factored `batch` steps at a time in one
    factorisation of the replicated pattern (chordal.probe_factors) instead of one after the other."""
    op = chordal.completion if which == "p" else chordal.cholesky
    first = True
    more = (lambda v: v > smin) if strict else (lambda v: v >= smin)
    # (a pattern whose single trial fills the device takes its trials one after the other, as in the bisection of
    # chordalsolver_feas: eight side by side cost eight trials where the search usually needs two to five, and the replicated
    # pattern they run on is then never built)
    batched = _batched() and base.symb.blklen <= int(options.get("batched_linesearch_maxlen", 1 << 20))
    while more(s):
        if first or not batched:
            first = False
            T = base + d * s
            try:
                op(T)
                if accept is None or accept(T, s):
                    return s
            except ArithmeticError:
                pass
            s *= beta
            continue
        steps = []
        while len(steps) < batch and more(s):
            steps.append(s)
            s *= beta
        ok, fac = chordal.probe_factors(base, d, steps, which)
        for k, st_ in enumerate(steps):
            if ok[k] and (accept is None or accept(fac[k], st_)):
                return st_
    return None


def chordalsolver_esd(A, b, primalstart=None, dualstart=None, scaling="primal", kktsolver="chol", p=None):
    """Extended self-dual embedding solver for

         minimize   <C, X>            maximize   b'y
         subject to <A_i, X> = b_i    subject to sum_i y_i A_i + S = C
                    X in C_V                      S in K_V

    (C_V: PSD-completable matrices with pattern V, K_V: PSD matrices with pattern V).
    Returns the reference's result dictionary (solvers.py:2451-2467) with scipy/numpy values.
    """
    BETA, EXPON, STEP, MINSTEP = 0.7, 3.0, 0.99, 1e-12   # hard-coded in the reference (solvers.py:1384-1387)
    T0w, T0 = time.perf_counter(), time.process_time()
    DEBUG = _opt("debug", bool)
    MAXITERS = _opt("maxiters", int, 1)
    ABSTOL = _opt("abstol", float, 0.0)
    RELTOL = _opt("reltol", float, 0.0)
    FEASTOL = _opt("feastol", float, 0.0, strict_lo=True)
    REFINEMENT = _opt("refinement", int, 0)
    show_progress = _opt("show_progress", bool)
    DIMACS = _opt("dimacs", bool)
    KKTREF = _opt("esd_kkt_refinement", int, 0)
    DS_HESS = _opt("esd_ds_from_hessian", bool)
    if scaling not in ("primal", "dual"):
        raise ValueError("scaling must be 'primal' or 'dual'")
    if kktsolver not in ("chol", "qr"):
        raise ValueError("Unknown 'kktsolver'.")

    P = A if isinstance(A, _Problem) else _Problem(A, b, p)
    P.use_kktsolver(kktsolver)
    n, m, C, bv = P.n, P.m, P.C, P.b
    Amap, Aadj = P.Amap, P.Aadj
    dot = chordal.dot
    bmax = float(np.abs(P.bh).max()) if m else 0.0
    ii = int(np.abs(P.bh).argmax()) if m else 0

    X = P.from_sym(primalstart["x"]) if primalstart is not None else P.identity()
    if dualstart is not None:
        y = torch.as_tensor(np.asarray(dualstart["y"], dtype=np.float64).reshape(-1), device=P.dev).clone()
        S = P.from_sym(dualstart["s"])
    else:
        S = P.identity()
        y = torch.zeros(m, dtype=torch.float64, device=P.dev)
    tau, kappa = 1.0, 1.0
    resy0 = max(1.0, _nrm2(bv))
    resx0 = max(1.0, math.sqrt(dot(C, C)))
    status, step = "unknown", None
    pcost = dcost = gap = relgap = pres = dres = pinfres = dinfres = None
    st = {}   # per-iteration state shared by the closures below

    def hess(U, inv):
        chordal.hessian(st["L"], st["Y"], U, adj=None, inv=inv)

    def bres(sigma, dz=None):
        t = st["t"]
        rby = (1 - sigma) * st["ry"]
        rbx = st["rx"] * (1 - sigma)
        rbt = (1 - sigma) * st["rt"]
        if scaling == "primal":
            rbs = st["L"].copy()
            chordal.llt(rbs)
            rbs *= sigma / t
            rbs -= S
            rbk = -kappa + sigma / (t * tau)
        else:
            rbs = st["Y"] * (sigma / t)
            rbs -= X
            rbk = -tau + sigma / (t * kappa)
        if dz is not None:
            ddy, ddX, ddtau, ddS, ddkappa = dz
            rby = rby + bv * ddtau - Amap(ddX)
            rbx += Aadj(ddy)
            rbx += ddS
            rbx -= C * ddtau
            rbt += dot(C, ddX) - _dot(bv, ddy) + ddkappa
            if scaling == "primal":
                rbs -= ddS
                u = ddX.copy()
                hess(u, True)
                rbs -= u * (1.0 / t)
                rbk -= ddkappa + ddtau / (t * tau ** 2)
            else:
                rbs -= ddX
                u = ddS.copy()
                hess(u, False)
                rbs -= u * (1.0 / t)
                rbk -= ddtau + ddkappa / (t * kappa ** 2)
        return rby, rbx, rbt, rbs, rbk

    def tres(rbz):
        t = st["t"]
        if scaling == "primal":
            a = tau ** 2 * t * (rbz[2] + rbz[4])
            rtx = C * a
            rtx -= rbz[1]
            rtx -= rbz[3]
        else:
            a = rbz[4] + rbz[2] / (t * kappa ** 2)
            rtx = rbz[3].copy()
            hess(rtx, True)
            rtx *= -t
            rtx += C * a
            rtx -= rbz[1]
        return rtx, rbz[0] + a * bv

    def solve(bx, by):
        x, yy = bx.copy(), by.clone()
        st["f"](x, yy, st["kk"])
        for _ in range(KKTREF):
            r = x.copy()
            hess(r, True)
            r *= -st["kk"]
            r += Aadj(yy)
            r -= bx
            rr = Amap(x) - by
            st["f"](r, rr, st["kk"])
            x -= r
            yy = yy - rr
        if DEBUG:   # the reference's KKT residual check (solvers.py:1801-1811, 1962-1966)
            r = x.copy()
            hess(r, True)
            r *= -st["kk"]
            r += Aadj(yy)
            r -= bx
            print("   KKTsolver: %.2e %.2e" % (math.sqrt(max(dot(r, r), 0.0)) / max(math.sqrt(dot(bx, bx)), 1e-300),
                                              _nrm2(Amap(x) - by) / max(_nrm2(by), 1e-300)))
        return x, yy

    def newton_once(sigma, dz):
        t = st["t"]
        rbz = bres(sigma, dz)
        rtx, rty = tres(rbz)
        u1, u2 = solve(rtx, rty)
        den = (1.0 / (t * tau ** 2) if scaling == "primal" else t * kappa ** 2)
        gamma = (-_dot(bv, u2) + dot(C, u1)) / (den + _dot(bv, st["v2"]) - dot(C, st["v1"]))
        dy = u2 + gamma * st["v2"]
        dX = u1 + st["v1"] * gamma
        dkappa = -rbz[2] + _dot(bv, dy) - dot(C, dX)
        if bmax > 1e-5:
            dtau = (float(Amap(dX)[ii].item()) - float(rbz[0][ii].item())) / P.bh[ii]
        elif scaling == "primal":
            dtau = (rbz[4] - dkappa) * t * tau ** 2
        else:
            dtau = rbz[4] - dkappa / (t * kappa ** 2)
        if DS_HESS:
            # the reference's choice (solvers.py:2017-2022, 2082-2086): dS through the inverse Hessian
            if scaling == "primal":
                dS = dX * (-1.0 / t)
                hess(dS, True)
                dS += rbz[3]
            else:
                dS = rbz[3] - dX
                dS *= t
                hess(dS, True)
        else:
            # dS from the dual-feasibility row of the Newton system (the variant the reference keeps as a
            # comment at solvers.py:2014-2016): algebraically identical, but it keeps the dual residual
            # exact to rounding instead of amplifying the error of dX by cond(W) ~ 1/mu^2 near the optimum
            dS = Aadj(-dy)
            dS += C * dtau
            dS -= rbz[1]
        return dy, dX, dtau, dS, dkappa

    def newton(sigma):
        dy, dX, dtau, dS, dkappa = newton_once(sigma, None)
        for _ in range(REFINEMENT):
            e = newton_once(sigma, (dy, dX, dtau, dS, dkappa))
            dy = dy + e[0]
            dX += e[1]
            dtau += e[2]
            dS += e[3]
            dkappa += e[4]
        return dy, dX, dtau, dS, dkappa

    def newton_res(sigma, dy, dX, dtau, dS, dkappa):
        """Residuals of the 5-block Newton system (reference DEBUG check, solvers.py:1813-1841)."""
        t = st["t"]
        rbz = bres(sigma)
        r1 = _nrm2(Amap(dX) - dtau * bv - rbz[0])
        r2 = Aadj(-dy)
        r2 += C * dtau
        r2 -= dS
        r2 -= rbz[1]
        r2 = math.sqrt(max(dot(r2, r2), 0.0))
        r3 = abs(_dot(bv, dy) - dot(C, dX) - dkappa - rbz[2])
        if scaling == "primal":
            r4 = dX.copy()
            hess(r4, True)
            r4 *= 1.0 / t
            r4 += dS
            r4 -= rbz[3]
            r5 = abs(dtau / (t * tau ** 2) + dkappa - rbz[4])
        else:
            r4 = dS.copy()
            hess(r4, False)
            r4 *= 1.0 / t
            r4 += dX
            r4 -= rbz[3]
            r5 = abs(dkappa / (t * kappa ** 2) + dtau - rbz[4])
        r4 = math.sqrt(max(dot(r4, r4), 0.0))
        print("   Newton residuals: %.2e %.2e %.2e %.2e %.2e" % (r1, r2, r3, r4, r5))

    def linesearch(dX, dS, dtau, dkappa):
        s = 1.0
        while kappa + s * dkappa <= 0 or tau + s * dtau <= 0:
            s *= BETA
            if s < MINSTEP:
                return None
        s = _backtrack(X, dX, "p", s, BETA, MINSTEP)
        if s is None:
            return None
        return _backtrack(S, dS, "d", s, BETA, MINSTEP)

    if show_progress:
        print("smcp_amd: extended self-dual embedding, %s scaling (Cholesky), n=%d m=%d cliques=%d%s"
              % (scaling, n, m, P.symb.Nsn, "" if P.ischordal else " (chordal embedding)"))
        print("%3s %12s %12s %8s %8s %8s %8s %8s" % ("it", "pcost", "dcost", "gap", "pres", "dres", "k/t", "step"))

    it = 0
    for it in range(MAXITERS + 1):
        hry = Amap(X)
        ry = bv * tau - hry
        hrx = Aadj(y)
        hrx += S
        rx = hrx - C * tau
        cx = dot(C, X)
        by = _dot(bv, y)
        rt = kappa - by + cx
        resy = _nrm2(ry) / tau
        resx = math.sqrt(max(dot(rx, rx), 0.0)) / tau
        pres, dres = resy / resy0, resx / resx0
        pcost, dcost = cx / tau, by / tau
        gap = dot(X, S) / tau ** 2
        relgap = gap / -pcost if pcost < 0.0 else (gap / dcost if dcost > 0.0 else None)
        pinfres = math.sqrt(max(dot(hrx, hrx), 0.0)) / resx0 / by if by > 0.0 else None
        dinfres = _nrm2(hry) / resy0 / (-cx) if cx < 0.0 else None
        if isinstance(options.get("trace"), list):       # per-iteration record for tests (not in the reference)
            options["trace"].append((it, pcost, dcost, gap, pres, dres))
        if show_progress:
            print("%3d % .4e % .4e %.1e %.1e %.1e %.1e %s" % (it, pcost, dcost, gap, pres, dres, kappa / tau,
                                                               "" if step is None else "%.1e" % step))
        if dres <= FEASTOL and pres <= FEASTOL and (gap <= ABSTOL or (relgap is not None and relgap <= RELTOL)):
            status = "optimal"
            break
        if pinfres is not None and pinfres <= FEASTOL:
            status = "primal infeasibility"
            X = None
            pcost, dcost, gap, relgap, pres, dres, dinfres = None, 1, None, None, None, None, None
            break
        if dinfres is not None and dinfres <= FEASTOL:
            status = "dual infeasibility"
            y = S = None
            pcost, dcost, gap, relgap, pres, dres, pinfres = -1, None, None, None, None, None, None
            break
        if it == MAXITERS:
            status = "unknown"
            break
        t = (n + 1) / (gap * tau ** 2 + kappa * tau)
        try:
            if scaling == "primal":
                L = X.copy()
                chordal.completion(L)
                Y = X
            else:
                L = S.copy()
                Y = cspmatrix(S.symb, torch.empty_like(L.blkval))
                chordal.cholesky_projected_inverse(L, Y)     # solvers.py:2341-2361 as one library call
        except ArithmeticError:
            status = "unknown"
            break
        try:
            f = P.factor(L, Y)
        except ArithmeticError:
            status = "unknown"
            break
        st.update(L=L, Y=Y, t=t, f=f, kk=(1.0 / t if scaling == "primal" else t), ry=ry, rx=rx, rt=rt)
        st["v1"], st["v2"] = solve(C, bv)
        dy, dX, dtau, dS, dkappa = newton(0.0)
        if DEBUG:
            newton_res(0.0, dy, dX, dtau, dS, dkappa)
        step = linesearch(dX, dS, dtau, dkappa)
        if not step:
            status = "unknown"
            break
        Xt, St = X + dX * step, S + dS * step
        taut, kappat = tau + step * dtau, kappa + step * dkappa
        sigma = ((dot(Xt, St) + taut * kappat) / (gap * tau ** 2 + kappa * tau)) ** EXPON
        dy, dX, dtau, dS, dkappa = newton(sigma)
        if DEBUG:
            newton_res(sigma, dy, dX, dtau, dS, dkappa)
        step = linesearch(dX, dS, dtau, dkappa)
        if not step:
            status = "unknown"
            break
        X += dX * (STEP * step)
        y = y + (STEP * step) * dy
        S += dS * (STEP * step)
        tau += STEP * step * dtau
        kappa += STEP * step * dkappa
        if DEBUG:
            print("   mu=%.2e tau=%.2e kappa=%.2e sigma=%.2e" % (1.0 / t, tau, kappa, sigma))

    dimacs = None
    if DIMACS and X is not None and y is not None and S is not None:
        R = Aadj(y)
        R += S
        R *= 1.0 / tau
        R -= C
        dimacs = [_nrm2(Amap(X) / tau - bv) / (1 + bmax), 0.0,
                  math.sqrt(max(dot(R, R), 0.0)) / (1 + P.cmaxabs), 0.0,
                  (pcost - dcost) / (1 + abs(pcost) + abs(dcost)), gap / (1 + abs(pcost) + abs(dcost))]
    xs = ys = ss = None
    if X is not None:
        X *= 1.0 / tau
        xs = P.to_scipy(X)
    if y is not None:
        ys = (y / tau).cpu().numpy()
    if S is not None:
        S *= 1.0 / tau
        ss = P.to_scipy(S)
    if show_progress:
        print("status: %s, %d iterations, %.2f s" % (status, it, time.perf_counter() - T0w))
    return {"status": status, "x": xs, "y": ys, "s": ss, "primal objective": pcost, "dual objective": dcost,
            "gap": gap, "relative gap": relgap, "primal infeasibility": pres, "dual infeasibility": dres,
            "residual as primal infeasibility certificate": pinfres,
            "residual as dual infeasibility certificate": dinfres, "iterations": it,
            "cputime": time.process_time() - T0, "time": time.perf_counter() - T0w, "dimacs": dimacs}


def chordalsolver_feas(A, b, primalstart=None, dualstart=None, scaling="primal", kktsolver="chol", p=None):
    """Feasible-start barrier method (reference: solvers.py:49-1327; behaviour summarised in
    SURVEY.md App. D.1).  Either a strictly feasible primal start {'x': X0} (primal scaling), a dual
    start {'y': y0[, 's': S0]} (dual scaling), both, or neither (start heuristics, solvers.py:722-814).
    Each pass: scaling point -> kkt_chol factor -> centering step (with `refinement` rounds of
    kkt_res + solve) until the Newton decrement is below `delta`, then lifting + approximate
    tangent step with 8-step bisection line search, optional prediction/corrector, gap update."""
    T0w, T0 = time.perf_counter(), time.process_time()
    DEBUG = _opt("debug", bool)
    MAXITERS = _opt("maxiters", int, 1)
    ABSTOL = _opt("abstol", float, 0.0)
    RELTOL = _opt("reltol", float, 0.0)
    FEASTOL = _opt("feastol", float, 0.0, strict_lo=True)
    REFINEMENT = _opt("refinement", int, 0)
    show_progress = _opt("show_progress", bool)
    DIMACS = _opt("dimacs", bool)
    DELTA = _opt("delta", float, 0.0, 1.0, strict_lo=True)
    ALPHA = _opt("alpha", float, 0.0, 0.5, strict_lo=True)
    BETA = _opt("beta", float, 0.0, 1.0, strict_lo=True)
    MINSTEP = _opt("minstep", float, 0.0, strict_lo=True)
    BATCHED = _opt("batched_linesearch", bool) and (has_device() or getattr(chordal, "_probe_emulated", False))
    LIFTING = _opt("lifting", bool)
    EQUALSTEPS = _opt("equalsteps", bool)
    PREDICTION = _opt("prediction", bool)
    STEP = _opt("step", float, 0.0, 1.0, strict_lo=True)
    t = float(_opt("t0", float, 0.0, strict_lo=True))
    ETA = options.get("eta")
    if ETA is not None:                              # solvers.py:132-136
        if type(ETA) is not float or ETA <= 0.0:
            raise TypeError("options['eta'] must be a positive float")
        ETATOL = 0.10 * ETA
    if scaling not in ("primal", "dual"):
        raise ValueError("scaling must be 'primal' or 'dual'")
    if kktsolver not in ("chol", "qr"):
        raise ValueError("Unknown 'kktsolver'.")

    P = A if isinstance(A, _Problem) else _Problem(A, b, p)
    P.use_kktsolver(kktsolver)
    n, m, C, bv = P.n, P.m, P.C, P.b
    Amap, Aadj, dot = P.Amap, P.Aadj, chordal.dot
    # one trial factorisation of a pattern this large fills the device by itself: eight side by side take eight times as
    # long, and the bisection below then needs three trials per round where the 8-ary round spends eight
    ONE_AT_A_TIME = P.symb.blklen > int(options.get("batched_linesearch_maxlen", 1 << 20))
    resy0 = max(1.0, _nrm2(bv))
    resx0 = max(1.0, math.sqrt(dot(C, C)))
    st = {}

    def S_of(y):
        S = Aadj(-y)
        S += C
        return S

    def in_cone(M, which):
        Lt = M.copy()
        (chordal.completion if which == "p" else chordal.cholesky)(Lt)
        return Lt

    def factor(L, Y):
        st.update(L=L, Y=Y, f=P.factor(L, Y))

    def kk_of(tt):
        return 1.0 / tt if st["scaling"] == "primal" else tt

    def solve(bx, by, tt):
        x, yy = bx.copy(), by.clone()
        st["f"](x, yy, kk_of(tt))
        return x, yy

    def kkt_res(x, yy, bx, by, tt):
        r = x.copy()
        chordal.hessian(st["L"], st["Y"], r, adj=None, inv=True)
        r *= -kk_of(tt)
        r += Aadj(yy)
        r -= bx
        return r, Amap(x) - by

    def solve_refined(bx, by, tt):
        x, yy = solve(bx, by, tt)
        for _ in range(REFINEMENT):
            r1, r2 = kkt_res(x, yy, bx, by, tt)
            dx_, dy_ = solve(r1, r2, tt)
            x -= dx_
            yy = yy - dy_
        return x, yy

    def ntdecr_primal(dx):
        du = dx.copy()
        chordal.hessian(st["L"], st["Y"], du, adj=True, inv=True)
        return math.sqrt(max(dot(du, du), 0.0))

    def ntdecr_dual(dy):
        du = Aadj(dy)
        chordal.hessian(st["L"], st["Y"], du, adj=False, inv=False)
        return math.sqrt(max(dot(du, du), 0.0))

    def bisect(base_, d, which, a):
        """Largest step g in [MINSTEP, 1] (to a resolution of 1/256 or finer) with base_ + g d inside the cone.
        The reference bisects with eight trial factorisations one after the other (solvers.py:615-689); with
        options['batched_linesearch'] the same interval is narrowed by an 8-ary search whose eight trial
        factorisations per round run concurrently on the device (chordal.probe_cone): three rounds, 1/512."""
        lo, hi, g_ = MINSTEP, 1.0, MINSTEP
        if BATCHED:
            KP = 8
            try:                               # the full step first, on its own: one factorisation instead of a round
                in_cone(base_ + d, which)
                return a
            except ArithmeticError:
                pass
            for _ in range(3):
                pts = [lo + (hi - lo) * (k + 1) / KP for k in range(KP)]
                if ONE_AT_A_TIME:
                    # hi is known to lie outside the cone (the full step, or a failed trial of the round before), and
                    # the feasible steps form an interval: the last feasible one of pts[0..6] by bisection, three
                    # trial factorisations one after the other instead of eight side by side
                    ia, ib = -1, KP - 1
                    while ib - ia > 1:
                        mid = (ia + ib) // 2
                        try:
                            in_cone(base_ + d * pts[mid], which)
                            ia = mid
                        except ArithmeticError:
                            ib = mid
                    kmax = ia
                else:
                    ok = chordal.probe_cone(base_, d, pts, which)
                    kmax = max([k for k in range(KP) if ok[k]], default=-1)
                if kmax >= 0:
                    lo = g_ = pts[kmax]
                if kmax == KP - 1:
                    break                      # the whole interval is feasible
                hi = pts[kmax + 1]
            return a * g_
        for _ in range(8):
            g = 0.5 * (lo + hi)
            try:
                in_cone(base_ + d * g, which)
                lo = g_ = g
            except ArithmeticError:
                hi = g
                g_ = lo
        return a * g_

    def linesearch_omega(X, dx, S, ds, eta):
        """Bisection on the step that keeps Omega(X, S) = phi_p(X) + phi_d(S) + n log(<X, S> / n) + n within ETATOL of
        eta (options['eta']; solvers.py:386-394, 662-689): one common primal/dual step."""
        lo, hi, g_ = MINSTEP, 1.0, None
        for _ in range(8):
            g = 0.5 * (lo + hi)
            Xt, St = X + dx * g, S + ds * g
            try:
                Lt = in_cone(Xt, "p")                # Cholesky factor of the inverse of the completion of Xt
                Lst = in_cone(St, "d")
                gapt = dot(St, Xt)
                g_ = g
                Ot = 2.0 * chordal.logdiagsum(Lt) - 2.0 * chordal.logdiagsum(Lst) + n * math.log(gapt / n)
                if Ot - eta > ETATOL:
                    hi = g
                elif Ot - eta < -ETATOL:
                    lo = g
                else:
                    break
            except (ArithmeticError, ValueError):    # outside a cone (or a non-positive gap)
                hi = g
                g_ = None
        return g_ if g_ else lo

    def _last_step(smin, strict):
        g = last = 1.0
        while (g > smin) if strict else (g >= smin):
            last = g
            g *= BETA
        return last, g

    def backtrack_primal(X, dx, tt, ntd):
        """Damped centering step on X (solvers.py:928-939)."""
        logdetL, tdcdx = chordal.logdiagsum(st["L"]), tt * dot(C, dx)

        def accept(Lt, gam):
            return gam * (tdcdx + gam * ALPHA * ntd ** 2) < 2 * (logdetL - chordal.logdiagsum(Lt))

        gam = _backtrack(X, dx, "p", 1.0, BETA, MINSTEP, accept, strict=True)
        if gam is None:                      # steps exhausted: the reference leaves the loop with its last trial
            last, gam = _last_step(MINSTEP, True)
            return X + dx * last, gam
        return X + dx * gam, gam

    def backtrack_dual(y, dy, tt, ntd):
        logdetL, ddyb = chordal.logdiagsum(st["L"]), -tt * _dot(dy, bv)

        def accept(Lt, gam):
            return gam * (ddyb + gam * ALPHA * ntd ** 2) < 2 * (chordal.logdiagsum(Lt) - logdetL)

        gam = _backtrack(S_of(y), Aadj(-dy), "d", 1.0, BETA, MINSTEP, accept, strict=True)
        last = gam
        if gam is None:
            last, gam = _last_step(MINSTEP, True)
        yt = y + last * dy
        return yt, S_of(yt), gam

    def step_into_cone(base_, d, which):
        """gam = 1, BETA, BETA^2, ... until base_ + gam d is inside the cone (or gam < 1e-14: the last trial is kept)."""
        gam = _backtrack(base_, d, which, 1.0, BETA, 1e-14)
        return gam if gam is not None else _last_step(1e-14, False)[0]

    # ---- starting points (solvers.py:691-814) ---------------------------------------------
    X = y = S = None
    if primalstart is not None:
        X = P.from_sym(primalstart["x"])
        if _nrm2(bv - Amap(X)) / resy0 > 1e-8:
            raise ValueError("infeasible primal starting point")
        try:
            in_cone(X, "p")
        except ArithmeticError:
            raise ValueError("infeasible primal starting point")
    if dualstart is not None:
        y = torch.as_tensor(np.asarray(dualstart["y"], dtype=np.float64).reshape(-1), device=P.dev).clone()
        if "s" in dualstart and dualstart["s"] is not None:
            S = P.from_sym(dualstart["s"])
            r = S_of(y) - S
            if math.sqrt(max(dot(r, r), 0.0)) / resx0 > 1e-8:
                raise ValueError("infeasible dual starting point")
        else:
            S = S_of(y)
        try:
            in_cone(S, "d")
        except ArithmeticError:
            raise ValueError("infeasible dual starting point")
    if primalstart is None and dualstart is None:
        st["scaling"] = "primal"
        Xt = P.identity()
        factor(in_cone(Xt, "p"), Xt)
        zero = cspmatrix(P.symb, torch.zeros_like(C.blkval))
        X0, _ = solve(zero, bv, 1.0)
        try:
            in_cone(X0, "p")
            X = X0
        except ArithmeticError:
            trA = Amap(Xt)
            Xb, _ = solve(zero, trA, 1.0)
            Xb -= Xt
            for sgn in (1.0, -1.0):
                try:
                    D = Xb * sgn
                    in_cone(D, "p")
                    gam = 2.0
                    while gam < 1e12:
                        try:
                            cand = X0 + D * gam
                            in_cone(cand, "p")
                            X = cand
                            break
                        except ArithmeticError:
                            gam *= 2
                    if X is not None:
                        break
                except ArithmeticError:
                    continue
        _, y0 = solve(C, torch.zeros_like(bv), 1.0)
        S0 = S_of(y0)
        try:
            in_cone(S0, "d")
            y, S = y0, S0
        except ArithmeticError:
            _, yh = solve(P.identity() * -1.0, torch.zeros_like(bv), 1.0)
            try:
                in_cone(Aadj(-yh), "d")
                for _ in range(200):
                    S0 = S_of(yh)
                    try:
                        in_cone(S0, "d")
                        y, S = yh, S0
                        break
                    except ArithmeticError:
                        yh = yh * 1.4
            except ArithmeticError:
                pass
    if X is None and y is None:
        raise ValueError("could not find a feasible starting point (solve Phase I problem instead)")
    if X is None and scaling == "primal":
        scaling = "dual"
    elif S is None and scaling == "dual":
        scaling = "primal"
    st["scaling"] = scaling

    if show_progress:
        print("smcp_amd: feasible-start barrier method, %s scaling (Cholesky), n=%d m=%d cliques=%d"
              % (scaling, n, m, P.symb.Nsn))
    gap = n / t
    pres = dres = pcost = dcost = relgap = None
    CENTER, status, it = True, "unknown", 0
    dxL = dyL = Shat = None
    for it in range(1, MAXITERS + 2):
        if scaling == "primal":
            pres = _nrm2(bv - Amap(X)) / resy0
            pcost = dot(C, X)
        else:
            r = S_of(y) - S
            dres = math.sqrt(max(dot(r, r), 0.0)) / resx0
            dcost = _dot(bv, y)
        relgap = (gap / -pcost if pcost is not None and pcost < 0.0 else
                  (gap / dcost if dcost is not None and dcost > 0.0 else None))
        if it == MAXITERS + 1:
            status = "unknown"
            break
        if dres is not None and pres is not None and pres < FEASTOL and dres < FEASTOL and \
                (gap < ABSTOL or (relgap is not None and relgap < RELTOL)):
            status = "optimal"
            break
        try:
            if scaling == "primal":
                L = in_cone(X, "p")
                Y = X.copy()
            else:
                L = in_cone(S, "d")
                Y = L.copy()
                chordal.projected_inverse(Y)
            factor(L, Y)
        except ArithmeticError:
            status = "unknown"
            break
        stype = "c"
        if CENTER:
            if scaling == "primal":
                Shat = L.copy()
                chordal.llt(Shat)
                bx = C - Shat * (1.0 / t)
                dx, lam = solve_refined(bx, bv - Amap(X), t)
                ntd = ntdecr_primal(dx)
                if ntd > DELTA:
                    if ntd >= 1.0:
                        X, _g = backtrack_primal(X, dx, t, ntd)
                    else:
                        X += dx
                else:
                    if LIFTING:
                        X -= dx
                        dxL = dx
                    else:
                        X += dx
                    y = lam
                    S = S_of(y)
                    CENTER = False
            else:
                nu, dy = solve_refined(S * -1.0, bv, t)
                ntd = ntdecr_dual(dy)
                if ntd > DELTA:
                    if ntd >= 1.0:
                        y, S, _g = backtrack_dual(y, dy, t, ntd)
                    else:
                        y = y + dy
                        S = S_of(y)
                else:
                    if LIFTING:
                        y = y - dy
                        dyL = dy
                    else:
                        y = y + dy
                    S = S_of(y)
                    X = nu
                    CENTER = False
        if not CENTER:
            stype = "a"
            dx, dy = solve_refined(S, bv - Amap(X), t)
            ds = Aadj(-dy)
            if ETA is not None:                      # stay in the Omega-neighbourhood (solvers.py:1046-1054)
                gam = linesearch_omega(X, dx, S, ds, ETA)
                pstep = dstep = gam
            else:
                pstep, dstep = bisect(X, dx, "p", STEP), bisect(S, ds, "d", STEP)
                if EQUALSTEPS:
                    pstep = dstep = min(pstep, dstep)
            Xt = X + dx * pstep
            yt = y + dstep * dy
            St = S_of(yt)
            if ETA is not None or not PREDICTION:
                X, y, S = Xt, yt, St
            else:
                gapt = dot(Xt, St)
                if LIFTING:
                    if scaling == "primal":
                        X += dxL
                    else:
                        y = y + dyL
                        S = S_of(y)
                t = n / gapt
                if scaling == "primal":
                    bx = S - Shat * (1.0 / t)
                else:
                    bx = X.copy()
                    chordal.hessian(L, Y, bx, adj=None, inv=True)
                    bx *= t
                    bx -= S
                dx, dy = solve_refined(bx, bv - Amap(X), t)
                if scaling == "primal":
                    ntd = ntdecr_primal(dx)
                    if ntd >= 1.0:
                        X, _g = backtrack_primal(X, dx, t, ntd)
                    else:
                        X += dx
                    gam = step_into_cone(S_of(y), Aadj(-dy), "d")
                    y = y + gam * dy
                    S = S_of(y)
                else:
                    ntd = ntdecr_dual(dy)
                    if ntd >= 1.0:
                        y, S, _g = backtrack_dual(y, dy, t, ntd)
                    else:
                        y = y + dy
                        S = S_of(y)
                    X = X + dx * step_into_cone(X, dx, "p")
            try:
                in_cone(X, "p")
                in_cone(S, "d")
            except ArithmeticError:
                status = "unknown"
                break
            gapt = dot(X, S)
            gap = min(n / t, gapt)
            t = n / gap
            pres = _nrm2(Amap(X) - bv) / resy0
            r = Aadj(y)
            r += S
            r -= C
            dres = math.sqrt(max(dot(r, r), 0.0)) / resx0
            pcost, dcost = dot(C, X), _dot(bv, y)
            CENTER = True
        if show_progress:
            print("%3d %s %s %s gap %.1e pres %s dres %s" % (
                it, stype, "% .4e" % pcost if pcost is not None else "      -     ",
                "% .4e" % dcost if dcost is not None else "      -     ", gap if gap is not None else n / t,
                "%.1e" % pres if pres is not None else "-", "%.1e" % dres if dres is not None else "-"))

    dimacs = None
    bmax = float(np.abs(P.bh).max()) if m else 0.0
    if DIMACS and X is not None and y is not None and S is not None and pcost is not None and dcost is not None:
        R = Aadj(y)
        R += S
        R -= C
        dimacs = [_nrm2(Amap(X) - bv) / (1 + bmax), 0.0, math.sqrt(max(dot(R, R), 0.0)) / (1 + P.cmaxabs), 0.0,
                  (pcost - dcost) / (1 + abs(pcost) + abs(dcost)), dot(X, S) / (1 + abs(pcost) + abs(dcost))]
    if show_progress:
        print("status: %s, %d iterations, %.2f s" % (status, it - 1, time.perf_counter() - T0w))
    return {"status": status, "x": P.to_scipy(X) if X is not None else None,
            "y": y.cpu().numpy() if y is not None else None, "s": P.to_scipy(S) if S is not None else None,
            "primal objective": pcost, "dual objective": dcost, "gap": gap, "relative gap": relgap,
            "primal infeasibility": pres, "dual infeasibility": dres, "iterations": it - 1,
            "cputime": time.process_time() - T0, "time": time.perf_counter() - T0w, "dimacs": dimacs}


# ---------------------------------------------------------------------------------------------
# CVXOPT-style front ends (solvers.py:2470-2699)
# ---------------------------------------------------------------------------------------------
def _embed_columns(cols, dims):
    """Map each column v (length Nl + sum(Nq) + sum(Ns^2)) of [h G] to vec of the block-diagonal
    matrix: LP part -> diagonal, SOC part -> arrow with the head in the LAST row (solvers.py:2512-2520),
    's' part -> lower triangle of the ns x ns block (2523-2529)."""
    Nl = int(dims.get("l", 0) or 0)
    Nq = [int(q) for q in dims.get("q", [])]
    Ns = [int(s) for s in dims.get("s", [])]
    n = Nl + sum(Nq) + sum(Ns)
    R, Cc, V = [], [], []
    cols = sp.csc_matrix(cols)
    for k in range(cols.shape[1]):
        v = np.asarray(cols[:, k].todense()).reshape(-1)
        I, J, W = [], [], []
        ptr, off = 0, 0
        for i in range(Nl):
            if v[i] != 0.0:
                I.append(i); J.append(i); W.append(v[i])
        ptr, off = Nl, Nl
        for nq in Nq:
            u0, u1 = v[ptr], v[ptr + 1:ptr + nq]
            if u0 != 0.0:
                for d in range(nq):
                    I.append(off + d); J.append(off + d); W.append(u0)
            for j in np.nonzero(u1)[0]:
                I.append(off + nq - 1); J.append(off + j); W.append(u1[j])
            ptr += nq
            off += nq
        for ns in Ns:
            blk = v[ptr:ptr + ns * ns].reshape((ns, ns), order="F")
            ii, jj = np.nonzero(np.tril(blk))
            for a, c in zip(ii, jj):
                I.append(off + a); J.append(off + c); W.append(blk[a, c])
            ptr += ns * ns
            off += ns
        I, J, W = np.asarray(I, dtype=np.int64), np.asarray(J, dtype=np.int64), np.asarray(W, dtype=np.float64)
        R.append(I + n * J)
        Cc.append(np.full(len(I), k, dtype=np.int64))
        V.append(W)
    A = sp.csc_matrix((np.concatenate(V), (np.concatenate(R), np.concatenate(Cc))), shape=(n * n, cols.shape[1]))
    return A, n, Nl, Nq, Ns


def conelp(c, G, h, dims=None, kktsolver="chol"):
    """minimize c'x s.t. Gx + s = h, s in K (cone given by dims = {'l','q','s'}); solved as the dual
    of a block-diagonal SDP through chordalsolver_esd (solvers.py:2470-2535).  Returns the reference's
    dictionary with 'x' (primal), 's' (slack) and 'z' (dual) as dense numpy vectors."""
    c = np.asarray(c, dtype=np.float64).reshape(-1)
    hh = np.asarray(h, dtype=np.float64).reshape(-1, 1)
    G = sp.csc_matrix(G)
    if dims is None:
        dims = {"l": G.shape[0], "q": [], "s": []}
    A, n, Nl, Nq, Ns = _embed_columns(sp.hstack([sp.csc_matrix(hh), G]), dims)
    sol = chordalsolver_esd(A, -c, kktsolver=kktsolver)
    X, S = sol["x"], sol["s"]

    def unpack(M, dual):
        if M is None:
            return None
        M = np.asarray(M.todense())
        out = [np.diag(M)[:Nl]]
        N = Nl
        for q in Nq:
            B = M[N:N + q, N:N + q]
            if dual:   # z0 = trace, z1 = 2 * last row (solvers.py:2551-2560)
                out.append(np.concatenate([[np.trace(B)], 2.0 * B[q - 1, :q - 1]]))
            else:      # s0 = corner, s1 = last row (solvers.py:2581-2583)
                out.append(np.concatenate([[B[q - 1, q - 1]], B[q - 1, :q - 1]]))
            N += q
        for s_ in Ns:
            out.append(M[N:N + s_, N:N + s_].reshape(-1, order="F"))
            N += s_
        return np.concatenate(out)

    sol["x"] = sol.pop("y")
    sol["z"] = unpack(X, True)
    sol["s"] = unpack(S, False)
    return sol


def lp(c, G, h, kktsolver="chol"):
    """minimize c'x s.t. Gx <= h (solvers.py:2600-2607)."""
    G = sp.csc_matrix(G)
    return conelp(c, G, h, {"l": G.shape[0], "q": [], "s": []}, kktsolver=kktsolver)


def _split_cone_parts(sol, Nl, sizes, tail_z, tail_s, square):
    """Replaces sol['z'] / sol['s'] by the per-cone pieces the reference's socp / sdp return
    (solvers.py:2626-2648, 2672-2697): 'zl', 'sl' (None without a linear part) and one entry per cone."""
    z, s_ = sol.pop("z"), sol.pop("s")
    if Nl:
        sol["zl"] = None if z is None else z[:Nl]
        sol["sl"] = None if s_ is None else s_[:Nl]
    else:
        sol["zl"] = sol["sl"] = None
    N = Nl
    zs, ss = [], []
    for ns in sizes:
        ln = ns * ns if square else ns
        for v, out in ((z, zs), (s_, ss)):
            if v is None:
                out.append(None)
            else:
                u = v[N:N + ln]
                out.append(u.reshape((ns, ns), order="F") if square else u)
        N += ln
    sol[tail_z], sol[tail_s] = zs, ss
    return sol


def socp(c, Gl=None, hl=None, Gq=None, hq=None, kktsolver="chol"):
    """Second-order cone program front end (solvers.py:2608-2650): returns 'x', 'zl', 'sl' and the lists 'zq', 'sq'
    in place of 'z', 's'.  The reference's dims['l'].append on an int is not reproduced."""
    if Gq is None or hq is None:
        raise ValueError("'Gq' and 'hq' cannot be zero")
    Gs, hs, dims = [], [], {"l": 0, "q": [], "s": []}
    if Gl is not None and hl is not None:
        Gl = sp.csc_matrix(Gl)
        Gs.append(Gl); hs.append(np.asarray(hl, dtype=np.float64).reshape(-1)); dims["l"] = Gl.shape[0]
    for Gk, hk in zip(Gq, hq):
        Gk = sp.csc_matrix(Gk)
        Gs.append(Gk); hs.append(np.asarray(hk, dtype=np.float64).reshape(-1)); dims["q"].append(Gk.shape[0])
    sol = conelp(c, sp.vstack(Gs), np.concatenate(hs), dims, kktsolver=kktsolver)
    return _split_cone_parts(sol, dims["l"], dims["q"], "zq", "sq", square=False)


def sdp(c, Gl=None, hl=None, Gs=None, hs=None, kktsolver="chol"):
    """SDP front end: Gs[k] has ns^2 rows (column-major vec), hs[k] is ns x ns (solvers.py:2651-2699); returns 'x',
    'zl', 'sl' and the lists 'zs', 'ss' of ns x ns matrices in place of 'z', 's'."""
    if Gs is None or hs is None:
        raise ValueError("'Gs' and 'hs' cannot be zero")
    Gall, hall, dims = [], [], {"l": 0, "q": [], "s": []}
    if Gl is not None and hl is not None:
        Gl = sp.csc_matrix(Gl)
        Gall.append(Gl); hall.append(np.asarray(hl, dtype=np.float64).reshape(-1)); dims["l"] = Gl.shape[0]
    for Gk, hk in zip(Gs, hs):
        Gk = sp.csc_matrix(Gk)
        ns = int(round(math.sqrt(Gk.shape[0])))
        Gall.append(Gk); hall.append(np.asarray(hk, dtype=np.float64).reshape(-1, order="F")); dims["s"].append(ns)
    sol = conelp(c, sp.vstack(Gall), np.concatenate(hall), dims, kktsolver=kktsolver)
    return _split_cone_parts(sol, dims["l"], dims["s"], "zs", "ss", square=True)

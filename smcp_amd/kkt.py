"""Device KKT layer: Amap / Aadj / kkt_chol (Schur build + factor) / solve_ closure.

Python-side mirror of the closures at src/python/solvers.py:369-386 and 477-541; all
arithmetic is done by the C-ABI entry points ``kkt_*`` (include/smcp_amd.h).
"""
import ctypes
import os
import weakref

import numpy as np
import torch

from . import _lib
from .chordal import _chk, _ensure
from .cspmatrix import cspmatrix, _stream, sync_cache


def _empty(n, dev):
    """Uninitialised fp64 device buffer -- under SMCP_POISON=1 (smcp_amd/csrc/switches.hpp) filled with 4.5e150, so that a
    read of a never-written entry shows in the results instead of meeting whatever the allocator left there."""
    t = torch.empty(n, dtype=torch.float64, device=dev)
    if os.environ.get("SMCP_POISON") == "1":
        t.fill_(4.5e150)
    return t


def column_range(m, rank, world):
    """Contiguous block of Schur-complement columns owned by `rank` (balanced to within one column)."""
    return (m * rank) // world, (m * (rank + 1)) // world


def _backend_is_gloo(group):
    import torch.distributed as dist
    return dist.get_backend(group) == "gloo"


def _all_reduce(t, group):
    """SUM all-reduce; device tensors are staged through the host when the backend is gloo (CPU tests,
    or two ranks sharing one GPU), RCCL takes them directly."""
    import torch.distributed as dist
    if t.is_cuda and _backend_is_gloo(group):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def _all_gather(out_list, t, group):
    import torch.distributed as dist
    if t.is_cuda and _backend_is_gloo(group):
        hs = [torch.empty(o.shape, dtype=o.dtype) for o in out_list]
        dist.all_gather(hs, t.cpu(), group=group)
        for o, h in zip(out_list, hs):
            o.copy_(h)
    else:
        dist.all_gather(out_list, t, group=group)


def _all_gather_into(recv, send, group):
    """One fused all-gather: recv (world * len(send)) <- every rank's send, both preallocated 1-D tensors."""
    import torch.distributed as dist
    if send.is_cuda and _backend_is_gloo(group):
        h = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_gather_into_tensor(h, send.cpu(), group=group)
        recv.copy_(h)
    else:
        dist.all_gather_into_tensor(recv, send, group=group)


def _all_to_all(recv, send, group):
    """One all-to-all with equal splits: recv (world * w) <- every rank's slice `me` of its send (world * w)."""
    import torch.distributed as dist
    if send.is_cuda and _backend_is_gloo(group):
        h = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(h, send.cpu(), group=group)
        recv.copy_(h)
    else:
        dist.all_to_all_single(recv, send, group=group)


def _gather_to(dst, recv_list, send, group):
    """Gather `send` of every rank on rank dst (recv_list: preallocated tensors there, None elsewhere)."""
    import torch.distributed as dist
    me = dist.get_rank(group)
    if send.is_cuda and _backend_is_gloo(group):
        hs = [torch.empty(send.shape, dtype=send.dtype) for _ in recv_list] if me == dst else None
        dist.gather(send.cpu(), hs, dst=dist.get_global_rank(group, dst) if group is not None else dst, group=group)
        if me == dst:
            for o, h in zip(recv_list, hs):
                o.copy_(h)
    else:
        dist.gather(send, recv_list if me == dst else None, dst=dist.get_global_rank(group, dst) if group is not None else dst, group=group)


class ShardedSchur:
    """Multi-GPU assembly of the Schur complement.

    Two sharding modes share this host logic (and are exercised on CPU with gloo + the oracle):

    * subtree sharding (default when a partition has been set with ``set_partition``): the clique tree
      is cut into subtrees owned by single ranks plus a small replicated top.  Gram formulation: every
      rank sweeps ALL m constraints over its own subtrees, the packed update blocks of the subtree roots
      are exchanged (one all-gather per chunk of right-hand sides), the top is swept redundantly, each
      rank forms the partial Gram matrix of its blkval ranges and ONE all-reduce completes H.
    * column sharding (fallback): each rank builds a column range of H with the reference formulation
      and one all-reduce of H completes it.

    Subclasses provide the compute: _columns, _potrf, and for the subtree mode _gram_prepare, _gram_sweep,
    _exchange_pack, _exchange_unpack, _gram_accumulate, _apply_partition."""

    partition = None
    force_sharded = False    # tests: take the sharded routes (and their collectives) with a group of ONE rank as well
    # Round 5: the TOP of the tree sharded by constraint.  With top_by_constraint the root blocks of the subtree roots travel in an
    # all-to-all instead of an all-gather -- rank q receives them for its share J_q of the constraints only (1 / N of the volume)
    # -- and sweeps the top for J_q only (1 / N of the replicated top's Schur phases); the top's panels of the shares are then
    # gathered on rank 0, which accumulates the top's block of H as before.  One collective more per Schur complement (the gather),
    # and the second Hessian of solve_ exchanges like the first (no rank holds every constraint's blocks any more).
    # Default (set by _install_partition): on from four ranks (emulated N = 4 / 8: 433 / 480 against 424 / 461 solves/s with the
    # replicated top, profiles/r05_shard_step_emul.json; at N = 2 the gather of the top panels costs what the top sweep saves);
    # SMCP_SHARD_TOP=constraint / replicated decides for every N.
    top_by_constraint = os.environ.get("SMCP_SHARD_TOP", "replicated") == "constraint"

    def set_partition(self, group):
        """Cut the tree for the ranks of `group` (deterministic: every rank computes the same cut)."""
        import torch.distributed as dist
        from .shard import subtree_partition
        P = self._install_partition(dist.get_world_size(group), dist.get_rank(group))
        if self.top_by_constraint and not self._by_share_collectives_work(group):
            self.top_by_constraint = False
        return P

    def _by_share_collectives_work(self, group):
        """The top by constraint needs an all-to-all and a gather where the replicated top needs an all-gather and an all-reduce
        only.  No multi-rank RCCL job has ever run in this build's environment: one tiny instance of each is tried when the
        partition is set, the ranks agree on the outcome through an all-reduce (which every route needs anyway), and a backend
        that refuses either sends ALL ranks to the replicated top instead of failing the first Schur complement."""
        import torch.distributed as dist
        world, rank = self._world(group)
        if world == 1:
            return True
        ok = 1.0
        try:
            send = torch.arange(world, dtype=torch.float64, device=self.dev) + 10.0 * rank
            recv = torch.empty(world, dtype=torch.float64, device=self.dev)
            _all_to_all(recv, send, group)
            got = [torch.empty(1, dtype=torch.float64, device=self.dev) for _ in range(world)] if rank == 0 else None
            _gather_to(0, got, send[:1].clone(), group)
            want = torch.arange(world, dtype=torch.float64) * 10.0 + rank
            if not torch.equal(recv.cpu(), want) or (rank == 0 and [float(g) for g in got] != [10.0 * r for r in range(world)]):
                ok = 0.0
        except Exception as e:                               # (an unsupported collective raises on every rank alike)
            import sys
            print("smcp_amd: top-by-constraint collectives unavailable (%s): replicated top" % (repr(e)[:120],), file=sys.stderr)
            ok = 0.0
        flag = torch.tensor([ok], dtype=torch.float64, device=self.dev)
        _all_reduce(flag, group)                             # (sum: every rank must have succeeded)
        return float(flag) > world - 0.5

    def _install_partition(self, world, rank):
        from .shard import subtree_partition
        P = self.partition = subtree_partition(self.symb, world)
        if "SMCP_SHARD_TOP" not in os.environ and "top_by_constraint" not in self.__dict__:
            self.top_by_constraint = world >= 4
        self._apply_partition(P, rank)
        # blkval positions this rank accounts for in sums over the whole matrix (the top counts once, on rank 0)
        mask = np.zeros(self.symb.blklen)
        for a, b in list(P.ranges_by_rank[rank]) + (list(P.top_ranges) if rank == 0 else []):
            mask[a:b] = 1.0
        self._own_mask = torch.from_numpy(mask).to(self.dev)
        self.__dict__.pop("_spair", None)
        self.__dict__.pop("_xchg", None)
        return P

    def _world(self, group):
        import torch.distributed as dist
        if group is None or not dist.is_initialized():
            return 1, 0
        return dist.get_world_size(group), dist.get_rank(group)

    def _exchange_plan(self, group, nrhs):
        world, rank = self._world(group)
        P = self.partition
        bufs = self.__dict__.setdefault("_xchg", {})
        if ("sizes", nrhs) not in bufs:
            bufs[("sizes", nrhs)] = [self._exchange_size(P.roots_by_rank[r], nrhs) for r in range(world)]
        sizes = bufs[("sizes", nrhs)]
        return bufs, sizes, max(max(sizes), 1)

    def _exchange(self, group, nrhs, live=True, keep=None):
        """Boundary exchange of a leaves->root sweep: the packed update blocks of the subtree roots of every rank
        (nrhs right-hand sides) travel in ONE all-gather on buffers that are allocated once per width.
        keep: a key under which the gathered buffer stays untouched by later exchanges (the Schur sweeps' blocks are
        combined again by the second Hessian of solve_, _exchange_local)."""
        world, rank = self._world(group)
        P = self.partition
        bufs, sizes, width = self._exchange_plan(group, nrhs)
        key = width if keep is None else ("keep", keep, width)
        if key not in bufs:
            bufs[key] = (torch.zeros(width, dtype=torch.float64, device=self.dev), _empty(width * world, self.dev))
        send, recv = bufs[key]
        if sizes[rank] and live:
            self._exchange_pack(P.roots_by_rank[rank], nrhs, send[:sizes[rank]])
        _all_gather_into(recv, send, group)
        self.collectives += 1
        if live:       # a rank whose sweep has failed only keeps the collective matched
            self._exchange_unpack_all(P, rank, nrhs, recv, width, sizes)
        return recv, width

    def _exchange_by_share(self, group, j0, j1, live=True):
        """Boundary exchange of the Schur sweeps of the constraints j0 .. j1 - 1 with the top sharded by constraint: ONE all-to-all;
        returns this rank's share [lo, hi) of the chunk (the constraints whose top sweep it runs next)."""
        world, rank = self._world(group)
        P = self.partition
        bufs, sizes1, _ = self._exchange_plan(group, 1)
        shares = []
        for d in range(world):
            a, b = column_range(self.m, d, world)
            shares.append((max(j0, a), max(max(j0, a), min(j1, b))))
        wmax = max(max(sizes1), 1) * max(max(hi - lo for lo, hi in shares), 1)
        key = ("a2a", wmax)
        if key not in bufs:
            bufs[key] = (torch.zeros(wmax * world, dtype=torch.float64, device=self.dev), _empty(wmax * world, self.dev))
        send, recv = bufs[key]
        if live and sizes1[rank]:
            for d, (lo, hi) in enumerate(shares):
                if hi > lo:
                    self._exchange_pack_range(P.roots_by_rank[rank], lo - j0, hi - lo, send[d * wmax:d * wmax + sizes1[rank] * (hi - lo)])
        _all_to_all(recv, send, group)
        self.collectives += 1
        lo, hi = shares[rank]
        if live and hi > lo:
            self._exchange_unpack_share(P, lo, hi, recv, wmax)
        return lo, hi

    def _gather_top_panels(self, group, live=True):
        """The top's rows of the swept stack: every rank holds them for its own share of the constraints; rank 0, which accumulates
        the top's block of H, receives the others' (one gather)."""
        world, rank = self._world(group)
        P = self.partition
        toplen = sum(b - a for a, b in P.top_ranges)
        if world == 1 or toplen == 0:
            return
        bufs = self.__dict__.setdefault("_xchg", {})
        width = max(column_range(self.m, d, world)[1] - column_range(self.m, d, world)[0] for d in range(world)) * toplen
        key = ("top", width)
        if key not in bufs:
            bufs[key] = (torch.zeros(width, dtype=torch.float64, device=self.dev),
                         [_empty(width, self.dev) for _ in range(world)] if rank == 0 else None)
        send, recv = bufs[key]
        c0, c1 = column_range(self.m, rank, world)
        if live and c1 > c0 and rank != 0:
            off = 0
            for a, b in P.top_ranges:
                self._stack_rows(0, c0, c1, a, b, send[off:off + (c1 - c0) * (b - a)])
                off += (c1 - c0) * (b - a)
        _gather_to(0, recv, send, group)
        self.collectives += 1
        if live and rank == 0:
            for r in range(1, world):
                r0, r1 = column_range(self.m, r, world)
                off = 0
                for a, b in P.top_ranges:
                    if r1 > r0:
                        self._stack_rows(1, r0, r1, a, b, recv[r][off:off + (r1 - r0) * (b - a)])
                    off += (r1 - r0) * (b - a)

    # most doubles kept from the Schur sweeps for the collective-free exchange of solve_'s second Hessian
    KEEP_LIMIT = 1 << 28

    def _exchange_local(self, group, y, recv1, kept):
        """The boundary blocks of the second Hessian of solve_ WITHOUT a collective: its input is Aadj(y) - bx
        (solvers.py:528-531), so the other ranks' root blocks are sum_i y_i (blocks gathered for constraint i by the
        Schur sweeps: `kept`, the chunks of the build_schur this solve_ belongs to) - (blocks gathered for bx by the first
        Hessian of THIS solve_: `recv1`, handed over by the caller) -- both are on this rank already."""
        world, rank = self._world(group)
        P = self.partition
        bufs, sizes1, width1 = self._exchange_plan(group, 1)
        if ("local", width1) not in bufs:
            bufs[("local", width1)] = _empty(width1 * world, self.dev)
        out = bufs[("local", width1)]
        out.copy_(recv1)
        for n, (j0, j1, recv, width) in enumerate(kept):
            self._exchange_combine(P, rank, j1 - j0, y[j0:j1], recv, width, out, width1, 0 if n == 0 else 1)
        self._exchange_unpack_all(P, rank, 1, out, width1, sizes1)

    def _exchange_unpack_all(self, P, rank, nrhs, recv, width, sizes):
        for r in range(len(sizes)):
            if r != rank and sizes[r]:
                self._exchange_unpack(P.roots_by_rank[r], nrhs, recv[r * width:r * width + sizes[r]])

    collectives = 0          # collectives issued so far (tests and DESIGN.md section 6 count them per KKT solve)

    def build_schur(self, L, Y, group=None):
        world, rank = self._world(group)
        if world == 1 and not (self.force_sharded and self.partition is not None and group is not None):
            self._columns(L, Y, 0, self.m)
            return
        P = self.partition
        if P is None:
            self.H.zero_()
            if self._sparse_count() > 0:
                # column-sparse constraints (misc.SCMcolumn2, solvers.py:489-497): every rank takes a contiguous share of them
                # (trsm x 2 + SCMcolumn2 per chunk), rank 0 the Gram block of the swept ones as well; factors replicated
                self._scm_part(L, Y, rank, world)
            else:
                j0, j1 = column_range(self.m, rank, world)
                if j1 > j0:
                    self._columns(L, Y, j0, j1)
            _all_reduce(self.H, group)
            self.collectives += 1
            return
        # ---- subtree-sharded Gram path.  A rank that fails (its part of a deferred factorisation, or chol(Y_AA) in its
        # sweeps) keeps taking part in the collectives; a status word rides on H's all-reduce and every rank raises.
        pend = self.__dict__.pop("_pending_status", None)
        if pend is not None and pend[0] is not L:
            raise RuntimeError("factor_scaling(defer_status=True) must be followed by factor() on the pair it returned")
        err = [pend[1] if pend is not None else None]

        def guarded(f, *args):
            if err[0] is None:
                try:
                    f(*args)
                except ArithmeticError as e:
                    err[0] = e

        if self._sharded_pair(L, Y):
            guarded(self._gram_prepare_part)                             # (L, Y) came from factor_scaling
        else:
            guarded(self._gram_prepare, L, Y)
        step = self._gram_chunk()
        # the gathered root blocks of every chunk are kept (if they fit KEEP_LIMIT doubles): solve_'s second Hessian
        # forms its own boundary blocks from them instead of a third exchange
        _, sizes_m, width_m = self._exchange_plan(group, min(step, self.m))
        keep = (width_m * world * ((self.m + step - 1) // step) <= self.KEEP_LIMIT
                and os.environ.get("SMCP_SHARD_KEEP", "1") != "0")       # 0: the second Hessian of solve_ exchanges like the first
        self._kept = []
        self._kept_gen = self.__dict__.get("_kept_gen", 0) + 1          # the kept chunks belong to THIS Schur complement
        by_share = self.top_by_constraint
        if by_share:
            keep = False
        for n, j0 in enumerate(range(0, self.m, step)):
            j1 = min(self.m, j0 + step)
            guarded(self._gram_sweep, 1, j0, j1)                         # owned subtrees
            if by_share:
                lo, hi = self._exchange_by_share(group, j0, j1, live=err[0] is None)
                if hi > lo:
                    guarded(self._gram_sweep, 2, lo, hi)                 # the top, for this rank's share of the chunk only
                continue
            recv, width = self._exchange(group, j1 - j0, live=err[0] is None, keep=n if keep else None)
            if keep:
                self._kept.append((j0, j1, recv, width))
            guarded(self._gram_sweep, 2, j0, j1)                         # replicated top
        if not keep:
            self._kept = None
        if by_share:
            self._gather_top_panels(group, live=err[0] is None)          # the shares' top panels -> rank 0 (one collective)
        ranges = list(P.ranges_by_rank[rank]) + (list(P.top_ranges) if rank == 0 else [])
        guarded(self._gram_accumulate, ranges)
        guarded(self._deferred_status)               # chordal.lazy_status: the one read-back of the step happens here
        self._Hbuf[-1] = 0.0 if err[0] is None else 1.0
        _all_reduce(self._Hbuf, group)
        self.collectives += 1
        if float(self._Hbuf[-1]) > 0:
            self.__dict__.pop("_spair", None)
            raise err[0] or ArithmeticError("not positive definite on another rank")

    def _deferred_status(self):
        pass

    # ---- sharded factorisation at a scaling point and the sharded solve_ (include/smcp_amd.h: *_part)
    def _sharded_pair(self, L, Y):
        sp = self.__dict__.get("_spair")
        return sp is not None and sp[0] is L and sp[1] is Y and sp[2] == (L.state(), Y.state())

    def factor_scaling(self, S, group=None, defer_status=False):
        """L = cholesky(S), Y = projected_inverse(L) (solvers.py:881-891) with every sweep sharded by subtree: the
        leaves->root factorisation runs on the owned cliques, ONE exchange hands the subtree roots' update blocks to
        every rank, the top is factored redundantly; the root->leaves inverse needs no communication.  Returns
        (L, Y), each valid on this rank's cliques and the top (other ranges keep S's values); `factor(L, Y, group)`
        recognises the pair and shards the Schur sweeps and solve_ the same way.
        A rank whose subtree is not positive definite must not leave the others in a collective: the failure is agreed
        on by an all-reduce of a status word before this returns -- or, with defer_status=True, together with H's
        all-reduce in the `factor(L, Y, group)` that must follow (one collective and one host read-back fewer per
        step; the ArithmeticError is then raised there, on every rank)."""
        from . import chordal
        world, rank = self._world(group)
        L = S.copy()
        if (world == 1 and not (self.force_sharded and group is not None)) or self.partition is None:
            chordal.cholesky(L)
            Y = L.copy()
            chordal.projected_inverse(Y)
            return L, Y
        err = None
        try:
            self._chol_part(L, 1)
        except ArithmeticError as e:     # keep the collectives matched: the failure is agreed on below
            err = e
        self._exchange(group, 1, live=err is None)
        try:
            self._chol_part(L, 2)
        except ArithmeticError as e:
            err = err or e
        Y = L.copy()
        if err is None:
            try:
                self._pinv_part(Y, 2)
                self._pinv_part(Y, 1)
                self._prepare_part(L, Y, 2)
                self._prepare_part(L, Y, 1)
            except ArithmeticError as e:
                err = e
        if defer_status:
            self.__dict__["_pending_status"] = (L, err)
            self.__dict__["_spair"] = (L, Y, (L.state(), Y.state()))
            return L, Y
        bad = torch.tensor([0.0 if err is None else 1.0], dtype=torch.float64, device=self.dev)
        _all_reduce(bad, group)
        self.collectives += 1
        if float(bad[0]) > 0:
            raise err or ArithmeticError("cholesky: not positive definite on another rank")
        self.__dict__["_spair"] = (L, Y, (L.state(), Y.state()))
        return L, Y

    def _solve_sharded(self, L, Y, bx, by, kk, group, complete, kept=None):
        """solve_ of kkt_chol (solvers.py:521-541) on a sharded factor: both Hessians run as owned sweep -> exchange ->
        top (up), top -> owned (down); Amap sums over the blkval ranges this rank accounts for and ONE all-reduce
        completes it; the m x m solve is replicated.  With complete=True one all-gather of the owned ranges fills x in on every
        rank, otherwise x is valid on the owned cliques and the top (what the next sharded sweep needs).
        kept: the gathered root blocks of the Schur sweeps of the factor() this solve_ came from (None: the second Hessian
        exchanges like the first)."""
        first = {}

        def W(U, y=None):
            self._hess_part(U, 1, 0)
            if y is not None and kept and "recv" in first:
                self._exchange_local(group, y, first["recv"], kept)   # no collective: Aadj(y) - bx is a combination of swept inputs
            else:
                # (a buffer of its own: no other exchange of the same width -- factor_scaling's, say -- can overwrite what the
                # second Hessian combines)
                first["recv"], _ = self._exchange(group, 1, keep="solve")
            self._hess_part(U, 2, 0)
            self._hess_part(U, 2, 1)
            self._hess_part(U, 1, 1)
        r1 = bx.copy()
        W(r1)
        r1.blkval.mul_(self._own_mask)
        yp = self.amap(r1)
        _all_reduce(yp, group)
        self.collectives += 1
        by.mul_(kk).add_(yp)
        self._potrs(by)
        x = self.aadj(by)
        bx.blkval.neg_().add_(x.blkval)
        bx.touched()
        W(bx, by)
        bx.blkval.mul_(1.0 / kk)
        if complete:
            self._complete_owned(bx, group)
        bx.touched()
        return bx, by

    def _complete_owned(self, X, group):
        """Fill X in on every rank: each rank contributes the blkval ranges of its subtrees (the top is already valid
        everywhere) through ONE all-gather -- |V| / world doubles sent per rank instead of an all-reduce of all of V."""
        world, rank = self._world(group)
        P = self.partition
        lens = [sum(b - a for a, b in P.ranges_by_rank[r]) for r in range(world)]
        width = max(max(lens), 1)
        bufs = self.__dict__.setdefault("_xchg", {})
        if ("own", width) not in bufs:
            bufs[("own", width)] = (torch.zeros(width, dtype=torch.float64, device=self.dev), _empty(width * world, self.dev))
        send, recv = bufs[("own", width)]
        o = 0
        for a, b in P.ranges_by_rank[rank]:
            send[o:o + b - a].copy_(X.blkval[a:b])
            o += b - a
        _all_gather_into(recv, send, group)
        self.collectives += 1
        for r in range(world):
            if r == rank:
                continue
            o = r * width
            for a, b in P.ranges_by_rank[r]:
                X.blkval[a:b].copy_(recv[o:o + b - a])
                o += b - a

    def _exchange_size(self, cliques, nrhs):
        na = np.diff(self.symb.rowptr) - np.diff(self.symb.snptr)
        return int(sum(int(na[k]) * (int(na[k]) + 1) // 2 for k in cliques) * nrhs)


def _forget_schur(symb_ref, Hbuf):
    symb = symb_ref()
    if symb is not None and getattr(symb, "_h", None):
        try:
            _lib.lib().kkt_schur_forget(symb.handle, Hbuf.data_ptr())
        except Exception:
            pass


class KKTSystem(ShardedSchur):
    """Holds the constraint matrices A_1..A_m (blkval coordinates) on the device."""

    def __init__(self, symb, cptr, cidx, cval, max_rhs=None, tnzcols=None):
        self.symb = symb
        self.m = len(cptr) - 1
        if max_rhs is None:
            # keep the per-chunk update workspace + constraint stack within ~8 GB
            per = 8 * (symb.updlen + 3 * symb.blklen)
            max_rhs = int(max(2, min(self.m, (8 << 30) // max(per, 1))))
        if symb._device is None or symb._max_rhs < max_rhs:
            if not torch.cuda.is_available():
                raise RuntimeError("smcp_amd needs an MI355X (HIP) device; there is no CPU fallback")
            symb.device_init(torch.cuda.current_device(), max_rhs)
        self._con = (np.ascontiguousarray(cptr, dtype=np.int64), np.ascontiguousarray(cidx, dtype=np.int64),
                     np.ascontiguousarray(cval, dtype=np.float64))
        self._tnzcols = 0.1 if tnzcols is None else float(tnzcols)      # the reference's default (solvers.py:31)
        self.dev = torch.device("cuda", symb._device)
        # H is a view of a buffer with one slot more: the agreed status of a deferred sharded factorisation rides on
        # H's all-reduce (factor_scaling(defer_status=True))
        self._Hbuf = torch.zeros(self.m * self.m + 1, dtype=torch.float64, device=self.dev)
        self.H = self._Hbuf[:self.m * self.m].view(self.m, self.m)
        # the context may remember H by address (a factorisation deferred under chordal.lazy_status, the cached inverses of
        # its diagonal blocks): it must forget it before the memory goes back to the allocator
        weakref.finalize(self, _forget_schur, weakref.ref(symb), self._Hbuf)
        self._install()

    # The constraint set (entry lists, classification, the swept stack / Q of kkt_qr) lives in the Symbolic's native
    # context, ONE set per context.  Each KKTSystem keeps its own host copy and re-installs it when another system
    # built on the same Symbolic has taken the context over, so two systems never run on each other's constraints;
    # state that cannot be re-created (the Q factor of factor_qr) is refused once the context has changed hands.
    def _install(self):
        lib = _lib.lib()
        cptr, cidx, cval = self._con
        _chk(lib.kkt_set_tnzcols(self.symb.handle, self._tnzcols), "kkt_set_tnzcols")
        _chk(lib.kkt_set_constraints(self.symb.handle, self.m, cptr.ctypes.data, cidx.ctypes.data, cval.ctypes.data),
             "kkt_set_constraints")
        d = self.symb.__dict__
        d["_kkt_epoch"] = d.get("_kkt_epoch", 0) + 1
        d["_kkt_owner"] = weakref.ref(self)
        self._epoch = d["_kkt_epoch"]

    def _own(self):
        o = self.symb.__dict__.get("_kkt_owner")
        if o is None or o() is not self:
            self._install()
            if self.partition is not None and getattr(self, "_part_rank", None) is not None:
                self._apply_partition(self.partition, self._part_rank)

    def amap(self, X):
        self._own()
        y = _empty(self.m, self.dev)
        _chk(_lib.lib().kkt_amap(self.symb.handle, X.blkval.data_ptr(), y.data_ptr(), _stream()), "kkt_amap")
        return y

    def aadj(self, y):
        self._own()
        X = cspmatrix(self.symb, _empty(self.symb.blklen, self.dev))
        X.touched()
        _chk(_lib.lib().kkt_aadj(self.symb.handle, y.data_ptr(), X.blkval.data_ptr(), _stream()), "kkt_aadj")
        return X

    def _columns(self, L, Y, j0, j1):
        self._own()
        sync_cache(self.symb, L, Y)
        _chk(_lib.lib().kkt_schur_columns(self.symb.handle, L.blkval.data_ptr(), Y.blkval.data_ptr(),
                                          self.H.data_ptr(), self.m, int(j0), int(j1), _stream()), "kkt_schur_columns")

    def _sparse_count(self):
        cnt = (ctypes.c_int64 * 2)()
        _chk(_lib.lib().kkt_constraint_classes(self.symb.handle, cnt), "kkt_constraint_classes")
        return int(cnt[1])

    def _scm_part(self, L, Y, part, nparts):
        self._own()
        sync_cache(self.symb, L, Y)
        _chk(_lib.lib().kkt_schur_gram_part(self.symb.handle, L.blkval.data_ptr(), Y.blkval.data_ptr(), self.H.data_ptr(), self.m,
                                            int(part), int(nparts), _stream()), "kkt_schur_gram_part")

    def _potrf(self):
        _chk(_lib.lib().dense_potrf(self.symb.handle, self.H.data_ptr(), self.m, self.m, _stream()), "dense_potrf")

    # ---- subtree-sharded Gram path (C-ABI: csp_set_partition, kkt_gram_*, csp_exchange_pack / unpack)
    def _apply_partition(self, P, rank):
        self._part_rank = int(rank)
        owner = np.ascontiguousarray(P.owner, dtype=np.int32)
        _chk(_lib.lib().csp_set_partition(self.symb.handle, owner.ctypes.data, int(rank)), "csp_set_partition")
        # the library derives the subtree roots from the owner array; the host logic sizes the collective from
        # P.roots_by_rank: the two must agree rank by rank
        world = len(P.roots_by_rank)
        sizes = np.zeros(world, dtype=np.int64)
        _chk(_lib.lib().csp_exchange_sizes(self.symb.handle, world, sizes.ctypes.data), "csp_exchange_sizes")
        want = [self._exchange_size(P.roots_by_rank[r], 1) for r in range(world)]
        if sizes.tolist() != want:
            raise RuntimeError("subtree roots of the partition disagree between host and library: %s vs %s" % (sizes.tolist(), want))

    def _gram_chunk(self):
        return int(self.symb._max_rhs)

    def _deferred_status(self):
        if self.symb.__dict__.get("_lazy_status"):
            from . import chordal
            chordal.check_status(self.symb)

    def _gram_prepare(self, L, Y):
        self._own()
        sync_cache(self.symb, L, Y)
        _chk(_lib.lib().kkt_gram_prepare(self.symb.handle, L.blkval.data_ptr(), Y.blkval.data_ptr(), _stream()),
             "kkt_gram_prepare")

    def _gram_sweep(self, which, j0, j1):
        _chk(_lib.lib().kkt_gram_sweep(self.symb.handle, int(which), int(j0), int(j1), _stream()), "kkt_gram_sweep")

    def _reprepare(self):
        """Another call on this Symbolic has rewritten the context's lk / yaa / fac buffers (SMCP_ESTALE): rebuild
        them from the sharded pair, top first."""
        L, Y = self._spair[0], self._spair[1]
        for which in (2, 1):
            _chk(_lib.lib().kkt_prepare_part(self.symb.handle, L.blkval.data_ptr(), Y.blkval.data_ptr(), which, 1,
                                             _stream()), "kkt_prepare_part")

    def _gram_prepare_part(self):
        self._own()
        rc = _lib.lib().kkt_gram_prepare_part(self.symb.handle, _stream())
        if rc == -5:
            self._reprepare()
            rc = _lib.lib().kkt_gram_prepare_part(self.symb.handle, _stream())
        _chk(rc, "kkt_gram_prepare_part")

    def _exchange_pack(self, cliques, nrhs, out):
        # this rank's subtree roots (the list csp_set_partition derived from the owner array: the same cliques)
        _chk(_lib.lib().csp_exchange_pack(self.symb.handle, int(nrhs), out.data_ptr(), _stream()), "csp_exchange_pack")

    def _exchange_pack_range(self, cliques, r0, nrhs, out):
        _chk(_lib.lib().csp_exchange_pack_range(self.symb.handle, int(r0), int(nrhs), out.data_ptr(), _stream()), "csp_exchange_pack_range")

    def _exchange_unpack_share(self, P, lo, hi, recv, width):
        _chk(_lib.lib().csp_exchange_unpack_all(self.symb.handle, int(hi - lo), recv.data_ptr(), int(width), _stream()), "csp_exchange_unpack_all")

    def _stack_rows(self, direction, j0, j1, a, b, buf):
        _chk(_lib.lib().kkt_stack_rows(self.symb.handle, int(direction), int(j0), int(j1), int(a), int(b), buf.data_ptr(), _stream()), "kkt_stack_rows")

    def _exchange_combine(self, P, rank, nrhs, y, recv, width, out, owidth, mode):
        _chk(_lib.lib().csp_exchange_combine(self.symb.handle, int(nrhs), y.data_ptr(), recv.data_ptr(), int(width),
                                             out.data_ptr(), int(owidth), int(mode), _stream()), "csp_exchange_combine")

    def _exchange_unpack_all(self, P, rank, nrhs, recv, width, sizes):
        _chk(_lib.lib().csp_exchange_unpack(self.symb.handle, int(nrhs), recv.data_ptr(), int(width), _stream()),
             "csp_exchange_unpack")

    # ---- sharded factorisation / solve sweeps (C-ABI: csp_cholesky_part, csp_projected_inverse_part,
    #      kkt_prepare_part, csp_hessian_sweep_part)
    def _chol_part(self, L, which):
        L.touched()
        _chk(_lib.lib().csp_cholesky_part(self.symb.handle, L.blkval.data_ptr(), int(which), _stream()), "cholesky")

    def _pinv_part(self, Y, which):
        Y.touched()
        _chk(_lib.lib().csp_projected_inverse_part(self.symb.handle, Y.blkval.data_ptr(), int(which), _stream()),
             "projected_inverse")

    def _prepare_part(self, L, Y, which):
        _chk(_lib.lib().kkt_prepare_part(self.symb.handle, L.blkval.data_ptr(), Y.blkval.data_ptr(), int(which), 0,
                                         _stream()), "kkt_prepare_part")

    def _hess_part(self, U, which, direction):
        U.touched()
        args = (self.symb.handle, U.blkval.data_ptr(), 1, self.symb.blklen, int(which), int(direction), _stream())
        rc = _lib.lib().csp_hessian_sweep_part(*args)
        if rc == -5 and which == 1 and direction == 0:      # only before the first sweep of a Hessian
            self._reprepare()
            rc = _lib.lib().csp_hessian_sweep_part(*args)
        _chk(rc, "csp_hessian_sweep_part")

    def _potrs(self, y):
        _chk(_lib.lib().dense_potrs(self.symb.handle, self.H.data_ptr(), self.m, self.m, y.data_ptr(), 1, self.m,
                                    _stream()), "dense_potrs")

    def _gram_accumulate(self, ranges):
        r = np.ascontiguousarray(np.asarray(ranges, dtype=np.int64).reshape(-1))
        _chk(_lib.lib().kkt_gram_accumulate(self.symb.handle, len(r) // 2, r.ctypes.data if len(r) else None,
                                            self.H.data_ptr(), self.m, _stream()), "kkt_gram_accumulate")

    def factor(self, L, Y, group=None):
        """kkt_chol(L, Y): builds (sharded over `group` if given) and factors the Schur complement;
        returns solve_(bx, by, kk).
        Under the deferred regime (chordal.lazy_status) a failure -- chol(Y_AA) inside the sweeps, potrf(H) -- is
        only LATCHED on the device: this call still returns a solve_, whose results are meaningless after a failure
        (safe: no index depends on a value), and the caller must read the verdict with chordal.check_status before
        it trusts them.  (The sharded build reads the latch itself, once, before H's all-reduce, because every rank
        has to agree on the outcome.)  The interior-point drivers run eagerly."""
        world, _ = self._world(group)
        if world == 1 and not (self.force_sharded and self.partition is not None and group is not None):
            # kkt_schur_factor: the Schur complement and its Cholesky factor in one call -- under chordal.lazy_status the
            # factorisation of H itself waits for the first solve_ and runs beside its first Hessian sweep (smcp_amd.h)
            self._own()
            sync_cache(self.symb, L, Y)
            _chk(_lib.lib().kkt_schur_factor(self.symb.handle, L.blkval.data_ptr(), Y.blkval.data_ptr(), self.H.data_ptr(), self.m,
                                             _stream()), "kkt_schur_factor")
        else:
            self.build_schur(L, Y, group)
            self._potrf()
        if self._sharded_pair(L, Y):
            # the chunks kept by THIS build_schur, with the generation they were gathered in: a solve_ that outlives a later
            # build_schur (whose sweeps reuse the buffers) falls back to the collective exchange for its second Hessian
            kept, gen = getattr(self, "_kept", None), self.__dict__.get("_kept_gen", 0)

            def solve_sharded(bx, by, kk, complete=True):
                """Overwrites bx (cspmatrix) with x and by (device vector) with y; sharded sweeps."""
                self._own()
                if not self._sharded_pair(L, Y):
                    raise RuntimeError("the sharded factor (L, Y) has been modified or replaced: factor again")
                live = kept if self.__dict__.get("_kept_gen", 0) == gen else None
                return self._solve_sharded(L, Y, bx, by, kk, group, complete, live)
            return solve_sharded

        def solve_(bx, by, kk):
            """Overwrites bx (cspmatrix) with x and by (device vector) with y."""
            self._own()
            sync_cache(self.symb, L, Y)
            bx.touched()
            _chk(_lib.lib().kkt_solve(self.symb.handle, L.blkval.data_ptr(), Y.blkval.data_ptr(),
                                      self.H.data_ptr(), self.m, float(kk), bx.blkval.data_ptr(),
                                      by.data_ptr(), _stream()), "kkt_solve")
            return bx, by

        return solve_

    def factor_qr(self, L, Y, group=None):
        """kkt_qr(L, Y) (solvers.py:413-475): QR factorisation of the stack of half-Hessian images of the
        constraints (csrc/kkt_qr.hip); returns solve_(bx, by, kk).  The system must have been created with
        tnzcols = 0 (every constraint swept).  Not sharded: one GPU."""
        if group is not None:
            import torch.distributed as dist
            if dist.is_initialized() and dist.get_world_size(group) > 1:
                raise NotImplementedError("kktsolver='qr' runs on one GPU (the Q factor is not sharded)")
        self._own()
        sync_cache(self.symb, L, Y)
        passes = ctypes.c_int64(0)
        shift = ctypes.c_double(0.0)
        _chk(_lib.lib().kkt_qr_factor(self.symb.handle, L.blkval.data_ptr(), Y.blkval.data_ptr(), ctypes.addressof(passes),
                                      ctypes.addressof(shift), _stream()), "kkt_qr_factor")
        self.qr_passes, self.qr_shift = passes.value, shift.value
        epoch = self.symb.__dict__["_kkt_epoch"]

        def solve_(bx, by, kk):
            """Overwrites bx (cspmatrix) with x and by (device vector) with y."""
            if self.symb.__dict__.get("_kkt_epoch") != epoch:
                raise RuntimeError("the Q factor of this kkt_qr factorisation is gone: another KKTSystem has used "
                                   "the same Symbolic since factor_qr; factor again")
            sync_cache(self.symb, L, Y)
            bx.touched()
            _chk(_lib.lib().kkt_qr_solve(self.symb.handle, L.blkval.data_ptr(), Y.blkval.data_ptr(), float(kk),
                                         bx.blkval.data_ptr(), by.data_ptr(), _stream()), "kkt_qr_solve")
            return bx, by

        return solve_

    def qr_inspect(self):
        """(R^T as a host array, Q^T Q as a device tensor) of the last factor_qr -- test hook."""
        Rt = np.zeros((self.m, self.m), order="F")
        G = torch.zeros((self.m, self.m), dtype=torch.float64, device=self.dev)
        _chk(_lib.lib().kkt_qr_inspect(self.symb.handle, Rt.ctypes.data, G.data_ptr(), _stream()), "kkt_qr_inspect")
        return Rt, G

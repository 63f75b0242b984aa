"""Device KKT layer: Amap / Aadj / kkt_chol (Schur build + factor) / solve_ closure.

Python-side mirror of the closures at src/python/solvers.py:369-386 and 477-541; all
arithmetic is done by the C-ABI entry points ``kkt_*`` (include/smcp_amd.h).
"""
import ctypes
import weakref

import numpy as np
import torch

from . import _lib
from .chordal import _chk, _ensure
from .cspmatrix import cspmatrix, _stream, sync_cache


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

    def set_partition(self, group):
        """Cut the tree for the ranks of `group` (deterministic: every rank computes the same cut)."""
        import torch.distributed as dist
        from .shard import subtree_partition
        world = dist.get_world_size(group)
        self.partition = subtree_partition(self.symb, world)
        self._apply_partition(self.partition, dist.get_rank(group))
        return self.partition

    def build_schur(self, L, Y, group=None):
        import torch.distributed as dist
        world = dist.get_world_size(group) if (group is not None and dist.is_initialized()) else 1
        if world == 1:
            self._columns(L, Y, 0, self.m)
            return
        rank = dist.get_rank(group)
        P = self.partition
        if P is None:
            j0, j1 = column_range(self.m, rank, world)
            self.H.zero_()
            if j1 > j0:
                self._columns(L, Y, j0, j1)
            _all_reduce(self.H, group)
            return
        # ---- subtree-sharded Gram path
        self._gram_prepare(L, Y)
        step = self._gram_chunk()
        for j0 in range(0, self.m, step):
            j1 = min(self.m, j0 + step)
            self._gram_sweep(1, j0, j1)                                  # owned subtrees
            mine = self._exchange_pack(P.roots_by_rank[rank], j1 - j0)   # 1-D tensor (may be empty)
            sizes = [self._exchange_size(P.roots_by_rank[r], j1 - j0) for r in range(world)]
            width = max(max(sizes), 1)
            send = torch.zeros(width, dtype=torch.float64, device=self.dev)
            send[:mine.numel()] = mine
            recv = [torch.empty(width, dtype=torch.float64, device=self.dev) for _ in range(world)]
            _all_gather(recv, send, group)
            for r in range(world):
                if r != rank and sizes[r]:
                    self._exchange_unpack(P.roots_by_rank[r], j1 - j0, recv[r][:sizes[r]])
            self._gram_sweep(2, j0, j1)                                  # replicated top
        ranges = list(P.ranges_by_rank[rank]) + (list(P.top_ranges) if rank == 0 else [])
        self._gram_accumulate(ranges)
        _all_reduce(self.H, group)

    def _exchange_size(self, cliques, nrhs):
        na = np.diff(self.symb.rowptr) - np.diff(self.symb.snptr)
        return int(sum(int(na[k]) * (int(na[k]) + 1) // 2 for k in cliques) * nrhs)


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
        self.H = torch.zeros((self.m, self.m), dtype=torch.float64, device=self.dev)
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
        y = torch.empty(self.m, dtype=torch.float64, device=self.dev)
        _chk(_lib.lib().kkt_amap(self.symb.handle, X.blkval.data_ptr(), y.data_ptr(), _stream()), "kkt_amap")
        return y

    def aadj(self, y):
        self._own()
        X = cspmatrix(self.symb, torch.empty(self.symb.blklen, dtype=torch.float64, device=self.dev))
        X.touched()
        _chk(_lib.lib().kkt_aadj(self.symb.handle, y.data_ptr(), X.blkval.data_ptr(), _stream()), "kkt_aadj")
        return X

    def _columns(self, L, Y, j0, j1):
        self._own()
        sync_cache(self.symb, L, Y)
        _chk(_lib.lib().kkt_schur_columns(self.symb.handle, L.blkval.data_ptr(), Y.blkval.data_ptr(),
                                          self.H.data_ptr(), self.m, int(j0), int(j1), _stream()), "kkt_schur_columns")

    def _potrf(self):
        _chk(_lib.lib().dense_potrf(self.symb.handle, self.H.data_ptr(), self.m, self.m, _stream()), "dense_potrf")

    # ---- subtree-sharded Gram path (C-ABI: csp_set_partition, kkt_gram_*, csp_exchange_copy)
    def _apply_partition(self, P, rank):
        self._part_rank = int(rank)
        owner = np.ascontiguousarray(P.owner, dtype=np.int32)
        _chk(_lib.lib().csp_set_partition(self.symb.handle, owner.ctypes.data, int(rank)), "csp_set_partition")

    def _gram_chunk(self):
        return int(self.symb._max_rhs)

    def _gram_prepare(self, L, Y):
        self._own()
        sync_cache(self.symb, L, Y)
        _chk(_lib.lib().kkt_gram_prepare(self.symb.handle, L.blkval.data_ptr(), Y.blkval.data_ptr(), _stream()),
             "kkt_gram_prepare")

    def _gram_sweep(self, which, j0, j1):
        _chk(_lib.lib().kkt_gram_sweep(self.symb.handle, int(which), int(j0), int(j1), _stream()), "kkt_gram_sweep")

    def _exchange_pack(self, cliques, nrhs):
        buf = torch.empty(self._exchange_size(cliques, nrhs), dtype=torch.float64, device=self.dev)
        if len(cliques):
            lst = np.ascontiguousarray(cliques, dtype=np.int64)
            _chk(_lib.lib().csp_exchange_copy(self.symb.handle, len(lst), lst.ctypes.data, int(nrhs), buf.data_ptr(), 0,
                                              _stream()), "csp_exchange_copy")
        return buf

    def _exchange_unpack(self, cliques, nrhs, buf):
        lst = np.ascontiguousarray(cliques, dtype=np.int64)
        buf = buf.contiguous()
        _chk(_lib.lib().csp_exchange_copy(self.symb.handle, len(lst), lst.ctypes.data, int(nrhs), buf.data_ptr(), 1,
                                          _stream()), "csp_exchange_copy")

    def _gram_accumulate(self, ranges):
        r = np.ascontiguousarray(np.asarray(ranges, dtype=np.int64).reshape(-1))
        _chk(_lib.lib().kkt_gram_accumulate(self.symb.handle, len(r) // 2, r.ctypes.data if len(r) else None,
                                            self.H.data_ptr(), self.m, _stream()), "kkt_gram_accumulate")

    def factor(self, L, Y, group=None):
        """kkt_chol(L, Y): builds (sharded over `group` if given) and factors the Schur complement;
        returns solve_(bx, by, kk)."""
        self.build_schur(L, Y, group)
        self._potrf()

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

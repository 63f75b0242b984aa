"""Device KKT layer: Amap / Aadj / kkt_chol (Schur build + factor) / solve_ closure.

Python-side mirror of the closures at src/python/solvers.py:369-386 and 477-541; all
arithmetic is done by the C-ABI entry points ``kkt_*`` (include/smcp_amd.h).
"""
import numpy as np
import torch

from . import _lib
from .chordal import _chk, _ensure
from .cspmatrix import cspmatrix, _stream


def column_range(m, rank, world):
    """Contiguous block of Schur-complement columns owned by `rank` (balanced to within one column)."""
    return (m * rank) // world, (m * (rank + 1)) // world


class ShardedSchur:
    """Multi-GPU assembly of the Schur complement: the m constraint sweeps are
    independent, so each rank builds its own column range and ONE all-reduce (RCCL over xGMI on GPUs,
    gloo in the CPU tests) of the m x m matrix completes H on every rank; potrf(H) and solve_ are
    replicated.  Subclasses provide _columns(L, Y, j0, j1) (fills self.H[:, j0:j1], column-major) and
    _potrf() (in-place Cholesky of self.H, raises ArithmeticError)."""

    def build_schur(self, L, Y, group=None):
        import torch.distributed as dist
        world = dist.get_world_size(group) if (group is not None and dist.is_initialized()) else 1
        if world == 1:
            self._columns(L, Y, 0, self.m)
            return
        rank = dist.get_rank(group)
        j0, j1 = column_range(self.m, rank, world)
        self.H.zero_()
        if j1 > j0:
            self._columns(L, Y, j0, j1)
        dist.all_reduce(self.H, op=dist.ReduceOp.SUM, group=group)


class KKTSystem(ShardedSchur):
    """Holds the constraint matrices A_1..A_m (blkval coordinates) on the device."""

    def __init__(self, symb, cptr, cidx, cval, max_rhs=None):
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
        cptr = np.ascontiguousarray(cptr, dtype=np.int64)
        cidx = np.ascontiguousarray(cidx, dtype=np.int64)
        cval = np.ascontiguousarray(cval, dtype=np.float64)
        _chk(_lib.lib().kkt_set_constraints(symb.handle, self.m, cptr.ctypes.data, cidx.ctypes.data,
                                            cval.ctypes.data), "kkt_set_constraints")
        self.dev = torch.device("cuda", symb._device)
        self.H = torch.zeros((self.m, self.m), dtype=torch.float64, device=self.dev)

    def amap(self, X):
        y = torch.empty(self.m, dtype=torch.float64, device=self.dev)
        _chk(_lib.lib().kkt_amap(self.symb.handle, X.blkval.data_ptr(), y.data_ptr(), _stream()), "kkt_amap")
        return y

    def aadj(self, y):
        X = cspmatrix(self.symb, torch.empty(self.symb.blklen, dtype=torch.float64, device=self.dev))
        _chk(_lib.lib().kkt_aadj(self.symb.handle, y.data_ptr(), X.blkval.data_ptr(), _stream()), "kkt_aadj")
        return X

    def _columns(self, L, Y, j0, j1):
        _chk(_lib.lib().kkt_schur_columns(self.symb.handle, L.blkval.data_ptr(), Y.blkval.data_ptr(),
                                          self.H.data_ptr(), self.m, int(j0), int(j1), _stream()), "kkt_schur_columns")

    def _potrf(self):
        _chk(_lib.lib().dense_potrf(self.symb.handle, self.H.data_ptr(), self.m, self.m, _stream()), "dense_potrf")

    def factor(self, L, Y, group=None):
        """kkt_chol(L, Y): builds (sharded over `group` if given) and factors the Schur complement;
        returns solve_(bx, by, kk)."""
        self.build_schur(L, Y, group)
        self._potrf()

        def solve_(bx, by, kk):
            """Overwrites bx (cspmatrix) with x and by (device vector) with y."""
            _chk(_lib.lib().kkt_solve(self.symb.handle, L.blkval.data_ptr(), Y.blkval.data_ptr(),
                                      self.H.data_ptr(), self.m, float(kk), bx.blkval.data_ptr(),
                                      by.data_ptr(), _stream()), "kkt_solve")
            return bx, by

        return solve_

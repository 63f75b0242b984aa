"""dense_potrf (lapack.potrf of the Schur complement, solvers.py:501) per call, one-launch dataflow route against the
per-step route (SMCP_FLOW=0 in the environment):  python3 tools/potrf_time.py [n ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smcp_amd import _lib, chordal, problems                      # noqa: E402
from smcp_amd.symbolic import Symbolic                            # noqa: E402

symb = Symbolic(problems.band_pattern(20, 2))
chordal._ensure(symb)
chordal.lazy_status(symb, True)
lib = _lib.lib()
for n in [int(v) for v in sys.argv[1:]] or [300, 576, 768, 1000, 2048, 4096]:
    rng = np.random.default_rng(n)
    M = rng.standard_normal((n, n))
    Hh = torch.from_numpy(M @ M.T + n * np.eye(n)).cuda()
    H = Hh.clone()
    reps = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for r in range(reps + 3):
        H.copy_(Hh)
        e0.record()
        lib.dense_potrf(symb.handle, H.data_ptr(), n, n, None)
        e1.record()
        e1.synchronize()
        if r >= 3:
            tot += e0.elapsed_time(e1)
    chordal.check_status(symb)
    ms = tot / reps
    print("n %5d  potrf %.3f ms  %.2f TFLOP/s  (%s)" % (n, ms, n ** 3 / 3 / ms / 1e9, "flow" if os.environ.get("SMCP_FLOW", "1") != "0" else "per-step"))

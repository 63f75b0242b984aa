import sys, os, ctypes
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from smcp_amd import _lib, chordal, problems
from smcp_amd.symbolic import Symbolic
symb = Symbolic(problems.band_pattern(50, 3)); chordal._ensure(symb)
lib = _lib.lib()
for n in (100, 37, 128, 300):
  for cond in (1e2, 1e8, 1e13):
    rng = np.random.default_rng(n)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    H = (Q * np.logspace(0, -np.log10(cond), n)) @ Q.T; H = (H + H.T) / 2
    xt = rng.standard_normal(n); b = H @ xt
    Hd = torch.from_numpy(H.copy()).cuda(); bd = torch.from_numpy(b.copy()).cuda()
    rc = lib.dense_potrf(symb.handle, Hd.data_ptr(), n, n, None)
    lib.dense_potrs(symb.handle, Hd.data_ptr(), n, n, bd.data_ptr(), 1, n, None)
    x = bd.cpu().numpy()
    Lr = np.linalg.cholesky(H)
    import scipy.linalg as sl
    xr = sl.cho_solve((Lr, True), b)
    print(os.environ.get("SMCP_POTRS_OLD","0"), os.environ.get("SMCP_POTRF_OLD","0"), n, "%.0e" % cond, rc,
          "resid gpu %.2e lapack %.2e" % (np.linalg.norm(H @ x - b) / np.linalg.norm(b), np.linalg.norm(H @ xr - b) / np.linalg.norm(b)),
          "fwd err gpu %.2e lapack %.2e" % (np.linalg.norm(x - xt) / np.linalg.norm(xt), np.linalg.norm(xr - xt) / np.linalg.norm(xt)),
          "factor err %.2e" % (np.linalg.norm(np.tril(Hd.cpu().numpy().T) - Lr) / np.linalg.norm(Lr)))

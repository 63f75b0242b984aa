import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
import scipy.linalg as sla
from test_gpu_parity import _kkt_qr_case, dev, host, orc, KKTSystem, rel
eps = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-9
m = 6
symb, S, msk, L, Yh, cptr, cidx, cval = _kkt_qr_case("nested_mid", m + 1, 41, density=0.05)
K0 = orc.KKT(S, cptr, cidx, cval)
dense = np.stack([K0.constraint(j) for j in range(m + 1)])
dense[1] = dense[0] + eps * dense[m]
cptr2, cidx2, cval2 = [0], [], []
for j in range(m):
    nz = np.flatnonzero(dense[j]); cidx2.extend(nz.tolist()); cval2.extend(dense[j][nz].tolist()); cptr2.append(len(cidx2))
cptr2, cidx2, cval2 = np.array(cptr2), np.array(cidx2), np.array(cval2)
K = orc.KKT(S, cptr2, cidx2, cval2)
F = K.qr_factor(L, Yh)
print("cond R", np.linalg.cond(F["R"]))
sys_ = KKTSystem(symb, cptr2, cidx2, cval2, max_rhs=4, tnzcols=0.0)
solve = sys_.factor_qr(dev(symb, L), dev(symb, Yh))
print("passes", sys_.qr_passes, "shift", sys_.qr_shift)
Rt, G = sys_.qr_inspect()
print("orth dev", np.abs(G.cpu().numpy() - np.eye(m)).max())
Rd = np.tril(Rt).T / np.sqrt(2.0)
Rr = F["R"] * np.sign(np.diag(F["R"]))[:, None]
print("R rel diff", np.abs(Rd - Rr).max() / np.abs(Rr).max(), "diag dev", np.diag(Rd), "diag ref", np.diag(Rr))
At = F["Q"] @ F["R"]
Qh = sla.solve_triangular(Rd, At.T, trans="T", lower=False).T
print("At Rdev^-1 orthonormal:", np.abs(Qh.T @ Qh - np.eye(m)).max())
rng = np.random.default_rng(42)
bx = rng.standard_normal(symb.blklen) * msk
by = np.zeros(m)
def res(x, y):
    r, rr = K.residual(L, Yh, x, y, bx, by, 1.0)
    return np.sqrt(orc.dot(S, r, r)) / max(1, np.sqrt(orc.dot(S, bx, bx))), np.linalg.norm(rr)
xr, yr = K.qr_solve(L, Yh, F, bx, by, 1.0)
print("oracle QR residuals", res(xr, yr), "|y|", np.abs(yr).max())
H = K.schur_factor(L, Yh, ncols=m)  # unfactored
try:
    Hc = H.copy(); orc.dense_potrf(Hc)
    xc, yc = K.solve(L, Yh, Hc, bx, by, 1.0)
    print("oracle chol residuals", res(xc, yc))
except Exception as e:
    print("oracle chol failed", e)
bxd, byd = dev(symb, bx), torch.from_numpy(by.copy()).cuda()
solve(bxd, byd, 1.0)
x, y = host(bxd) * msk, byd.cpu().numpy()
print("device QR residuals", res(x, y), "|y|", np.abs(y).max())
print("x rel diff vs oracle", rel(x[msk], xr[msk]), "y", y, yr)
try:
    s2 = sys_.factor(dev(symb, L), dev(symb, Yh))
    bxd, byd = dev(symb, bx), torch.from_numpy(by.copy()).cuda()
    s2(bxd, byd, 1.0)
    print("device chol residuals", res(host(bxd) * msk, byd.cpu().numpy()))
except Exception as e:
    print("device chol failed:", e)

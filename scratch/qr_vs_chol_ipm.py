import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from smcp_amd import base, solvers, kkt
solvers.options.update(show_progress=False, maxiters=100)
P = base.band_SDP(40, 12, 2, seed=13)
orig_qr = kkt.KKTSystem.factor_qr
calls = {"n": 0}
def wrapped(self, L, Y, group=None):
    from smcp_amd.cspmatrix import cspmatrix
    f_qr = orig_qr(self, L, Y, group)
    Rt, G = self.qr_inspect()
    orth = float((G - torch.eye(self.m, dtype=torch.float64, device=G.device)).abs().max())
    condR = np.linalg.cond(np.tril(Rt))
    passes, shift = self.qr_passes, self.qr_shift
    g = torch.Generator(device="cpu"); g.manual_seed(calls["n"])
    msk = torch.zeros(L.symb.blklen, dtype=torch.float64); msk[torch.from_numpy(L.symb.ccs_to_blk())] = 1.0
    bx0 = (torch.randn(L.symb.blklen, generator=g, dtype=torch.float64) * msk).cuda() * (0.0 if os.environ.get('ZERO_BX') else 1.0)
    by0 = torch.randn(self.m, generator=g, dtype=torch.float64).cuda()
    res = []
    for kk in (1.0, 1e-3, 1e-6):
        bx, by = cspmatrix(L.symb, bx0.clone()), by0.clone()
        f_qr(bx, by, kk)
        res.append((bx, by))
    f_chol = self.factor(L, Y)
    out = []
    for (bxq, byq), kk in zip(res, (1.0, 1e-3, 1e-6)):
        bx, by = cspmatrix(L.symb, bx0.clone()), by0.clone()
        f_chol(bx, by, kk)
        out.append("kk %.0e dx %.1e dy %.1e" % (kk, float((bxq.blkval - bx.blkval).norm() / bx.blkval.norm()), float((byq - by).norm() / by.norm())))
    calls["n"] += 1
    print("factor %3d condR %.1e passes %d shift %.0e orth %.1e | %s" % (calls["n"], condR, passes, shift, orth, " | ".join(out)), flush=True)
    return orig_qr(self, L, Y, group)
kkt.KKTSystem.factor_qr = wrapped
s = P.solve_feas(scaling="primal", kktsolver="qr")
print(s["status"], s["iterations"], s["primal objective"], s["gap"])

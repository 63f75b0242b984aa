import numpy as np, torch, sys
sys.path.insert(0, '.')
from tests.test_gpu_parity import *
from tests.test_gpu_parity import _launch_counts
for name, m, density in [("nested_mid", 12, 0.02), ("fam_max", 9, 0.01), ("fam_odd", 10, 0.03), ("nested", 8, 0.03)]:
    symb, S, A, msk = setup(name, 11)
    L = A.copy(); orc.cholesky(S, L); Yh = L.copy(); orc.projected_inverse(S, Yh)
    cptr, cidx, cval = problems.random_constraints(symb, m, density=density, seed=13)
    sys_ = KKTSystem(symb, cptr, cidx, cval, max_rhs=m, tnzcols=0.0)
    chordal.tune(symb, chordal.TUNE_LEAFGRAM, 2)
    Ld, Yd = dev(symb, L), dev(symb, Yh)
    counts = _launch_counts(symb, lambda: sys_.factor(Ld, Yd))
    print(name, symb.Nsn, len(cidx), counts)

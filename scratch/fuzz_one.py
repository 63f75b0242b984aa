"""One case of the randomised parity sweep, tag printed first: python scratch/fuzz_one.py <seed>"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import fuzz_parity
from smcp_amd.symbolic import Symbolic, amalgamate
seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
pat = fuzz_parity.pattern(rng, 0)
symb = Symbolic(pat)
if rng.random() < 0.4:
    emb = amalgamate(symb)
    if emb is not None:
        symb = Symbolic(emb[0], emb[1])
nn, na = symb.clique_sizes()
print("seed", seed, "n", symb.n, "nsn", symb.Nsn, "maxnn", symb.max_nn, "maxna", symb.max_na, "nlev", symb.nlev, flush=True)
big = [(int(a), int(b)) for a, b in zip(nn, na) if a > 16 or b > 64]
print("fronts beyond the small class:", big[:20], flush=True)
worst = fuzz_parity.run(1, seed, verbose=True)
print("OK", {k: "%.1e" % v[0] for k, v in worst.items()})

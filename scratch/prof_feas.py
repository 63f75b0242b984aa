import sys, os, time, cProfile, pstats
sys.path.insert(0, os.getcwd())
import numpy as np
from smcp_amd import base, solvers
solvers.options.update(show_progress=False)
P = base.band_SDP(200, 100, 3, seed=0)
ps, ds = {"x": P._X0}, {"y": P._y0, "s": P._S0}
P.solve_feas(scaling="dual", primalstart=ps, dualstart=ds)
pr = cProfile.Profile()
pr.enable()
sol = P.solve_feas(scaling="dual", primalstart=ps, dualstart=ds)
pr.disable()
print(sol["status"], sol["iterations"], sol["time"])
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)

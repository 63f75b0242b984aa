import sys, os, time, cProfile, pstats
sys.path.insert(0, os.getcwd())
from smcp_amd import base, problems, solvers
solvers.options.update(show_progress=False, maxiters=100)
P = base.pattern_SDP(problems.nested_block_arrow_pattern(), 100, density=0.005, seed=0)
kw = dict(scaling="dual", primalstart={"x": P._X0}, dualstart={"y": P._y0, "s": P._S0})
pr = cProfile.Profile(); pr.enable()
sol = P.solve_feas(**kw)
pr.disable()
print(sol["status"], sol["iterations"], sol["time"])
pstats.Stats(pr).sort_stats("tottime").print_stats(16)

import sys, os, io, contextlib
sys.path.insert(0, os.getcwd())
from smcp_amd import base, solvers
solvers.options.update(show_progress=True, maxiters=100)
P = base.band_SDP(40, 12, 2, seed=13)
logs = {}
for ks in ("chol", "qr"):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        P.solve_feas(scaling="primal", kktsolver=ks)
    logs[ks] = buf.getvalue().splitlines()
for a, b in zip(logs["chol"], logs["qr"]):
    print(("  " if a == b else "!!"), a[:75].ljust(75), "|", b[:75])
for extra in logs["qr"][len(logs["chol"]):]:
    print("++", " " * 75, "|", extra[:75])

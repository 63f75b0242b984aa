"""Long randomised parity sweep (GPU vs oracle).  Usage: python tools/fuzz_parity.py [ncases] [seed0]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import fuzz_parity
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
worst = fuzz_parity.run(ncases, seed0, verbose=True)
print("worst relative errors over %d cases:" % ncases)
for k, (v, tag) in sorted(worst.items()):
    print("  %-32s %.2e   %s" % (k, v, tag))
bad = {k: v for k, v in worst.items() if not v[0] <= 1e-7}
print("FAIL" if bad else "OK")
sys.exit(1 if bad else 0)

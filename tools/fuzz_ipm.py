"""Long interior-point fuzz (tests/fuzz_ipm.py): python3 tools/fuzz_ipm.py [ncases] [seed0] [first]"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import fuzz_ipm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
bad = fuzz_ipm.run(n, seed0, first, verbose=True)
print("cases", n, "mismatches", len(bad))
sys.exit(1 if bad else 0)

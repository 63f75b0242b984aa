"""Long runs of tests/fuzz_sharded.py: python tools/fuzz_sharded.py [ncases] [seed0] [world]"""
import os, runpy, sys
sys.argv[0] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "fuzz_sharded.py")
runpy.run_path(sys.argv[0], run_name="__main__")

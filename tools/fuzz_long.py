"""Long randomised parity run on the final code: python3 tools/fuzz_long.py [ncases] [seed0] -> gpurun_out/fuzz_long.json"""
import sys, os, json, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import fuzz_parity
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 90000
t0 = time.time()
worst = {}
for b in range(0, n, 20):
    w = fuzz_parity.run(min(20, n - b), seed0=seed0 + b)
    for k, (v, tag) in w.items():
        if k not in worst or v > worst[k][0]:
            worst[k] = (v, tag)
    print("cases %d..%d done, %.0f s, worst so far %.2e" % (b, b + 19, time.time() - t0, max(v for v, _ in worst.values())), flush=True)
json.dump({k: [v, str(t)] for k, (v, t) in worst.items()}, open("gpurun_out/fuzz_long.json", "w"), indent=1)
bad = {k: v for k, v in worst.items() if v[0] > 1e-9}
print("checks", len(worst), "bad", bad)
sys.exit(1 if bad else 0)

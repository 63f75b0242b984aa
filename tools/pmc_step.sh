#!/bin/bash
# Counter figures of ONE bench step of a workload, per kernel and summed:
#   [WORKLOAD=synth50k|dense4096|arrow|maxcut|synth50k_dense|synth50k_trace|band200] [ROUND=r05] bash tools/pmc_step.sh
#   -> gpurun_out/${ROUND}_pmc_${WORKLOAD}.json   (copy to profiles/; bench.py reads it for roofline.traffic and roofline.step)
# Three separate rocprofv3 --pmc passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes (HBM section): FETCH_SIZE (doubled:
# gfx950 counts 64 B per 128-B request of wide coalesced reads), WRITE_SIZE (as is; both in KB), SQ_INSTS_VALU_MFMA_MOPS_F64
# (x 512 = flops executed on the matrix pipe).  One step = the last period of the repeating launch sequence of the timed loop.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export ROUND=${ROUND:-r05} WORKLOAD=${WORKLOAD:-synth50k}
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU_MFMA_MOPS_F64; do
  rm -rf gpurun_out/pmc_$c
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_$c -o p -- python3 bench.py --workload $WORKLOAD --no-secondary --steps 3 --warmup 1 --no-cpu --no-profile --no-back-solve --no-check > gpurun_out/pmc_$c.log 2>&1 || { echo "pass $c failed"; tail -5 gpurun_out/pmc_$c.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, collections, json, os, sys
sys.path.insert(0, '.')
import bench
w, rnd = os.environ["WORKLOAD"], os.environ["ROUND"]
def short(n):
    return n.split('(')[0].replace('void ', '').replace('smcp::', '').replace('(anonymous namespace)::', '')
def one_step(counter):
    f = glob.glob('gpurun_out/pmc_%s/**/*counter_collection.csv' % counter, recursive=True)
    rows = [r for r in csv.DictReader(open(f[0])) if r['Counter_Name'] == counter]
    # one row per (dispatch, counter) -- sum over the rows of a dispatch (the tool may split by dimension)
    disp = collections.OrderedDict()
    for r in rows:
        d = int(r['Dispatch_Id'])
        e = disp.setdefault(d, [short(r['Kernel_Name']), 0.0])
        e[1] += float(r['Counter_Value'])
    seq = [disp[d] for d in sorted(disp)]
    names = [s[0] for s in seq]
    N = len(names)
    period, end = None, N
    def rep(e, p, exact):
        a, b, c = names[e - p:e], names[e - 2 * p:e - p], names[e - 3 * p:e - 2 * p]
        return (a == b == c) if exact else (sorted(a) == sorted(b) == sorted(c) and a[0] == b[0] == c[0])
    for exact in (True, False):
        for e in range(N, max(N - 600, 0), -1):
            for p in range(10, e // 3 + 1):
                if rep(e, p, exact):
                    period, end = p, e
                    break
            if period:
                break
        if period:
            break
    if period is None:
        raise SystemExit("no repeating launch sequence found in the %s pass" % counter)
    out = collections.OrderedDict()
    for name, v in seq[end - period:end]:
        e = out.setdefault(name, [0, 0.0])
        e[0] += 1; e[1] += v
    return out, period
F, pf = one_step('FETCH_SIZE')
W, pw = one_step('WRITE_SIZE')
M, pm = one_step('SQ_INSTS_VALU_MFMA_MOPS_F64')
ker = {}
for k in sorted(set(F) | set(W) | set(M)):
    n = max(F.get(k, [0])[0], W.get(k, [0])[0], M.get(k, [0])[0], 1)
    fk, wk, mk = F.get(k, [0, 0.0])[1], W.get(k, [0, 0.0])[1], M.get(k, [0, 0.0])[1]
    ker[k] = {"launches_per_step": n, "fetch_size_kb_per_launch": round(fk / n), "write_size_kb_per_launch": round(wk / n),
              "hbm_bytes_per_launch": int(1024 * (2 * fk + wk) / n), "mfma_flops_per_launch": int(512 * mk / n)}
tot_b = sum(v["hbm_bytes_per_launch"] * v["launches_per_step"] for v in ker.values())
tot_f = sum(v["mfma_flops_per_launch"] * v["launches_per_step"] for v in ker.values())
top = sorted(ker.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_per_step"])
label = bench.build_workload(w)[1] if hasattr(bench, "build_workload") else w
json.dump({"_comment": "per kernel and per step from rocprofv3 PMC counters, three separate passes (--pmc FETCH_SIZE ; WRITE_SIZE ; "
           "SQ_INSTS_VALU_MFMA_MOPS_F64) of `python3 bench.py --workload %s --no-secondary --steps 3 --warmup 1 --no-cpu --no-profile "
           "--no-back-solve --no-check` on MI355X; one step = the last period of the repeating launch sequence (%d / %d / %d launches in the "
           "three passes); FETCH_SIZE doubled per /opt/skills/guides/MI355X_MICROARCH.md (HBM section), WRITE_SIZE as is (KB); MFMA "
           "flops = MOPS x 512" % (w, pf, pw, pm),
           "workload_key": w, "csrc_sha256": bench.csrc_sha256(), "launches_per_step": pf,
           "step": {"hbm_bytes": tot_b, "mfma_flops": tot_f}, "kernels": dict(top)},
          open('gpurun_out/%s_pmc_%s.json' % (rnd, w), 'w'), indent=1)
print("step: %.3f GB counter traffic, %.3e executed MFMA flops, %d launches" % (tot_b / 1e9, tot_f, pf))
for k, v in top[:12]:
    print(k, v)
PY

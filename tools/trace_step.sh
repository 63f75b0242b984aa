# Kernel timeline of ONE step (names, grids, durations, gaps): [WORKLOAD=maxcut] bash tools/trace_step.sh -> gpurun_out/trace_step[_$WORKLOAD].txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
w=${WORKLOAD:-synth50k}
out=gpurun_out/trace_step
rm -rf $out; mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 bench.py --workload $w --no-secondary --steps 3 --warmup 1 --no-cpu --no-profile --no-back-solve --no-check > $out/run.log 2>&1
WORKLOAD=$w python3 - <<'PY'
import csv, glob, os
w = os.environ["WORKLOAD"]
f = glob.glob("gpurun_out/trace_step/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the timed steps repeat the same launch sequence: the last step = the shortest period p with three equal repetitions ending at
# some launch e near the end of the trace (the untimed checks of bench.py follow the timed loop).  Launches of side streams can
# swap places between steps: compare the sorted name multisets of the candidate windows as a fallback.
N = len(names)
period, end = None, N
def rep(e, p, exact):
    a, b, c = names[e - p:e], names[e - 2 * p:e - p], names[e - 3 * p:e - 2 * p]
    return (a == b == c) if exact else (sorted(a) == sorted(b) == sorted(c) and a[0] == b[0] == c[0])
for exact in (True, False):
    for e in range(N, max(N - 400, 0), -1):
        for p in range(20, e // 3):
            if rep(e, p, exact):
                period, end = p, e
                break
        if period:
            break
    if period:
        break
if period is None:
    period, end = min(N, 400), N
i0 = end - period
rows = rows[:end]
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
dst = "gpurun_out/trace_step.txt" if w == "synth50k" else "gpurun_out/trace_step_%s.txt" % w
with open(dst, "w") as o:
    for r in rows[i0:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("smcp::", "").replace("(anonymous namespace)::", "")
        o.write("%9.1f us  +%6.1f gap  %8.1f us  grid %-18s wg %-5s %s\n" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3,
                "%sx%sx%s" % (r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"]), r["Workgroup_Size_X"], n[:60]))
        prev_end = e
print(dst, period, "launches")
PY

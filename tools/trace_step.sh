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
# the timed steps repeat the same launch sequence: the last step = the shortest period the tail of the trace repeats with
N = len(names)
period = None
for p in range(8, N // 3):
    if names[N - p:] == names[N - 2 * p:N - p] and names[N - p:] == names[N - 3 * p:N - 2 * p]:
        period = p
        break
if period is None:
    period = min(N, 400)
i0 = N - period
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

# Kernel timeline of ONE synth50k step (names, grids, durations, gaps): bash tools/trace_step.sh -> gpurun_out/trace_step.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/trace_step
rm -rf $out; mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 bench.py --no-secondary --steps 2 --warmup 1 --no-cpu --no-profile --no-back-solve > $out/run.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace_step/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step = from the last k_chol_mfma<true> launch pair back ... simply: take the last 1/3 of launches after the last 'copyBuffer' burst
names = [r["Kernel_Name"] for r in rows]
# find the start of the last step: last occurrence of the first cholesky kernel of a step
starts = [i for i, n in enumerate(names) if "k_chol_mfma" in n]
# cholesky launches come in pairs per step (two LDS levels); the step starts a few launches before
i0 = starts[-2] - 2
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
with open("gpurun_out/trace_step.txt", "w") as o:
    for r in rows[i0:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("smcp::", "").replace("(anonymous namespace)::", "")
        o.write("%9.1f us  +%6.1f gap  %8.1f us  grid %-18s wg %-5s %s\n" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3,
                "%sx%sx%s" % (r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"]), r["Workgroup_Size_X"], n[:60]))
        prev_end = e
PY
tail -3 gpurun_out/trace_step.txt

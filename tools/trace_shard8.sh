# Kernel timeline of one emulated rank-0-of-8 step (tools/shard_step_emul.py --worlds 8): gpurun_out/trace_shard8.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/trace_shard8
rm -rf $out; mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 tools/shard_step_emul.py --mode subtree --worlds 8 --steps 3 > $out/run.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace_shard8/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
starts = [i for i, n in enumerate(names) if "k_chol_mfma" in n]
i0 = starts[-2] - 2
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
with open("gpurun_out/trace_shard8.txt", "w") as o:
    for r in rows[i0:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("smcp::", "").replace("(anonymous namespace)::", "")
        o.write("%9.1f us  +%6.1f gap  %8.1f us  grid %-18s wg %-5s %s\n" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3,
                "%sx%sx%s" % (r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"]), r["Workgroup_Size_X"], n[:60]))
        prev_end = e
PY
tail -2 gpurun_out/trace_shard8.txt

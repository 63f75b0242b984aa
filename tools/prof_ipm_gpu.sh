# GPU-side view of a whole interior-point run on synth50k: kernel time by name against the wall time of the run
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_ipm
rm -rf $out; mkdir -p $out
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o s -- python3 tools/ipm_synth50k.py > $out/run.log 2>&1
tail -2 $out/run.log
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_ipm/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.3f s over %d launches" % (tot / 1e9, sum(int(r["Calls"]) for r in rows)))
for r in rows[:22]:
    print("%7.1f ms %6d calls %6.1f us  %s" % (float(r["TotalDurationNs"]) / 1e6, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:90]))
PY

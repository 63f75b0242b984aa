# host (cProfile) and device (rocprofv3 kernel stats) view of the whole max-cut interior-point run (n = 1000, 5909 edges)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_ipm_mc
rm -rf $out; mkdir -p $out
python3 tools/maxcut.py 1000 5909 > $out/plain.log 2>&1; tail -2 $out/plain.log
python3 -c "
import cProfile, pstats, sys, runpy
sys.argv = ['tools/maxcut.py', '1000', '5909']
pr = cProfile.Profile(); pr.enable()
try: runpy.run_path('tools/maxcut.py', run_name='__main__')
finally:
    pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(30)
    pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
" > $out/host.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o s -- python3 tools/maxcut.py 1000 5909 > $out/run.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_ipm_mc/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.3f s over %d launches" % (tot / 1e9, sum(int(r["Calls"]) for r in rows)))
for r in rows[:25]:
    print("%7.1f ms %6d calls %6.1f us  %s" % (float(r["TotalDurationNs"]) / 1e6, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:90]))
PY

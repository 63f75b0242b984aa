# Everything profiles/r05_* is made of, on the code as it stands:  bash tools/profile_round5.sh   (one gpurun call, ~10 min)
#   gpurun_out/prof_r05/: r05_pmc_<workload>.json (tools/pmc_step.sh), r05_<workload>_kernel_stats.csv (rocprofv3 --kernel-trace --stats),
#   r05_step_timeline[_maxcut].txt (tools/trace_step.sh), r05_shard_step_emul.json (tools/shard_step_emul.py), r05_bench_full.json
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export ROUND=r05
out=gpurun_out/prof_r05
rm -rf $out; mkdir -p $out
for w in ${WORKLOADS:-synth50k maxcut dense4096 arrow synth50k_dense synth50k_trace band200}; do
  WORKLOAD=$w bash tools/pmc_step.sh > $out/pmc_$w.log 2>&1 && cp gpurun_out/r05_pmc_$w.json $out/ || { echo "pmc $w failed"; tail -5 $out/pmc_$w.log; }
  echo "pmc $w: $(grep '^step:' $out/pmc_$w.log)"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$w -o s -- python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu --no-secondary --no-back-solve > $out/bench_$w.json 2> $out/stats_$w.log \
    && cp $out/stats_$w/*kernel_stats.csv $out/r05_${w}_kernel_stats.csv
  rm -rf $out/stats_$w
done
bash tools/trace_step.sh > $out/trace.log 2>&1; cp gpurun_out/trace_step.txt $out/r05_step_timeline.txt
WORKLOAD=maxcut bash tools/trace_step.sh >> $out/trace.log 2>&1; cp gpurun_out/trace_step_maxcut.txt $out/r05_step_timeline_maxcut.txt
timeout 600 python3 tools/shard_step_emul.py --mode subtree --top both --steps 30 > $out/emul.log 2>&1; cp gpurun_out/shard_step_emul.json $out/r05_shard_step_emul.json; grep -c emulated $out/emul.log
# the full bench line LAST, with the PMC summaries of this very code in place (bench.py reads profiles/)
mkdir -p profiles && cp $out/r05_pmc_*.json profiles/
python3 bench.py --steps 20 --warmup 5 > $out/r05_bench_full.json 2> $out/bench_full.err; echo "bench rc=$?"; tail -c 300 $out/bench_full.err
python3 - <<'PY'
import json
r = json.loads(open("gpurun_out/prof_r05/r05_bench_full.json").read().strip().splitlines()[-1])
print(r["value"], r["ms_per_step"], r["roofline"]["kernel"], r["roofline"]["frac"], r["roofline"]["bound"], r["roofline"].get("traffic"), r["roofline"].get("step"))
print(r.get("stages")); print(r.get("back_solve", {}).get("ms"))
for k, v in r.get("secondary", {}).items():
    print(k, v.get("value"), v.get("ms_per_step"), (v.get("roofline") or {}).get("kernel"), (v.get("roofline") or {}).get("frac"), (v.get("roofline") or {}).get("traffic"), ((v.get("roofline") or {}).get("step") or {}).get("hbm", {}).get("frac"))
PY

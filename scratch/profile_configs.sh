# bench lines (with the CPU leg) and rocprofv3 kernel stats of configs 2 and 3: bash scratch/profile_configs.sh -> gpurun_out/prof_cfg/
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_cfg
rm -rf $out; mkdir -p $out
for w in arrow dense4096; do
  timeout 500 python3 bench.py --workload $w --steps 3 > $out/bench_$w.json 2> $out/bench_$w.err || exit 1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$w -o s -- python3 bench.py --workload $w --steps 3 --no-cpu > $out/stats_$w.log 2>&1 || exit 1
  cp $out/stats_$w/*kernel_stats.csv $out/kernel_stats_$w.csv
  tail -1 $out/bench_$w.json | cut -c1-250
done

# rocprofv3 kernel stats of configs 2, 3 and 4 (GPU legs only): bash tools/profile_configs.sh -> gpurun_out/prof_cfg/
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_cfg
rm -rf $out; mkdir -p $out
for w in ${WORKLOADS:-arrow dense4096 maxcut synth50k_dense}; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$w -o s -- python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu --no-secondary --no-back-solve > $out/bench_$w.json 2> $out/stats_$w.log || exit 1
  cp $out/stats_$w/*kernel_stats.csv $out/kernel_stats_$w.csv
  rm -rf $out/stats_$w
  tail -1 $out/bench_$w.json | cut -c1-200
done

# Round-3 profiles of the final code: bash scratch/profile_r03.sh  -> gpurun_out/prof_r03/ (copy the summaries to profiles/)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_r03
rm -rf $out; mkdir -p $out
python3 bench.py > $out/bench.json 2> $out/bench.err && \
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 bench.py --no-secondary --steps 5 --no-cpu --no-back-solve > $out/stats.log 2>&1 && \
cp $out/stats/*kernel_stats.csv $out/kernel_stats.csv && \
bash scratch/hbm_traffic.sh > $out/hbm.log 2>&1 && \
bash scratch/pmc_mfma.sh > $out/mfma.log 2>&1 && \
bash scratch/trace_step.sh > $out/trace.log 2>&1
cp gpurun_out/r03_hbm_traffic.json $out/ 2>/dev/null
cp gpurun_out/trace_step.txt $out/step_timeline.txt 2>/dev/null
tail -c 600 $out/bench.json; tail -8 $out/hbm.log; tail -6 $out/mfma.log

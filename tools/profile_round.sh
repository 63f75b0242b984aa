# Profiles of the headline configuration on the code as it stands:  ROUND=r04 bash tools/profile_round.sh
#   -> gpurun_out/prof_$ROUND/  {bench.json, kernel_stats.csv, ${ROUND}_hbm_traffic.json, mfma.log, step_timeline.txt}
# (copy the summaries to profiles/ under the names DESIGN.md section 4 quotes)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export ROUND=${ROUND:-r04}
out=gpurun_out/prof_$ROUND
rm -rf $out; mkdir -p $out
python3 bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err && \
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 bench.py --no-secondary --steps 5 --no-cpu --no-back-solve > $out/stats.log 2>&1 && \
cp $out/stats/*kernel_stats.csv $out/kernel_stats.csv && \
bash tools/hbm_traffic.sh > $out/hbm.log 2>&1 && \
bash tools/pmc_mfma.sh > $out/mfma.log 2>&1 && \
bash tools/trace_step.sh > $out/trace.log 2>&1
cp gpurun_out/${ROUND}_hbm_traffic.json $out/ 2>/dev/null
cp gpurun_out/trace_step.txt $out/step_timeline.txt 2>/dev/null
rm -rf $out/stats
tail -c 600 $out/bench.json; tail -8 $out/hbm.log; tail -6 $out/mfma.log

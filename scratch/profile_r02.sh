# Round-2 profiles of the final code: bash scratch/profile_r02.sh  -> gpurun_out/prof_r02/ (copy the summaries to profiles/)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_r02
rm -rf $out; mkdir -p $out
python3 bench.py --no-secondary > $out/bench.json 2> $out/bench.err && \
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 bench.py --no-secondary --steps 5 --no-cpu --no-back-solve > $out/stats.log 2>&1 && \
cp $out/stats/*kernel_stats.csv $out/kernel_stats.csv && \
bash scratch/hbm_traffic.sh > $out/hbm.log 2>&1 && \
bash scratch/pmc_mfma.sh > $out/mfma.log 2>&1
tail -1 $out/bench.json | cut -c1-300; tail -8 $out/hbm.log; tail -6 $out/mfma.log

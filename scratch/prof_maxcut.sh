cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/mc3 -o mc -- python3 scratch/maxcut.py 1000 5909 > gpurun_out/mc3.log 2>&1
grep -E "^status|time" gpurun_out/mc3.log | tail -2
f=$(find gpurun_out/mc3 -name "*kernel_stats.csv" | head -1)
head -22 $f | cut -c1-160

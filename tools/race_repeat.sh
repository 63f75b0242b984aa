# repeat the sharded delay-injection test N times with SMCP_FUZZ_RACE_SEEDS seeds each; print what moved when a run fails
cd $GRAFT_REPO_ROOT
for i in $(seq 1 ${N:-6}); do
  python3 -m pytest tests/test_gpu_distributed.py -x -q -k "sharded_step_under_delay" > gpurun_out/race_rep_$i.log 2>&1
  echo "run $i rc=$?"; grep -h "AssertionError: rank" gpurun_out/race_rep_$i.log | cut -c1-1200
done

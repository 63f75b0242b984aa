# how often does the two-rank full-size sharded step disagree with the single-rank step?  N runs per setting (a wrong result, not a fault)
cd $GRAFT_REPO_ROOT
N=${N:-10}
for setting in "" "SMCP_EVENT_SYSFENCE=1" "SMCP_FORK=0"; do
  bad=0
  for i in $(seq 1 $N); do
    env $setting python3 -m pytest tests/test_gpu_distributed.py -x -q -k "synth50k_full_size and gloo" > gpurun_out/flake_one.log 2>&1 || { bad=$((bad+1)); grep -h "differs from" gpurun_out/flake_one.log | cut -c1-600; }
  done
  echo "[$setting] failures $bad / $N"
done

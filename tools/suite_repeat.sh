# the whole GPU suite N times (no -x): which tests fail, and what the sharded delay-injection test says when it does
cd $GRAFT_REPO_ROOT
for i in $(seq 1 ${N:-3}); do
  python3 -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/suite_rep_$i.log 2>&1
  echo "suite run $i rc=$? $(tail -1 gpurun_out/suite_rep_$i.log)"; grep -h "^FAILED\|AssertionError: rank\|differs from" gpurun_out/suite_rep_$i.log | cut -c1-1500
done

# Race hunt:  bash tools/race_hunt.sh   (one gpurun call; SEEDS="1 2 3" N=20 to widen)
# Seeded delay injection (SMCP_RACE / csp_tune CSP_TUNE_RACE, smcp_amd/csrc/capi.hip: a random 5 .. 200 us spin kernel at the
# head and tail of every internal side-stream branch, behind every fork on the caller's stream and before one launch in
# eight) under
#   1. the in-process tests that loop over seeds themselves (single-rank step: lazy + eager; sharded synth50k step on two
#      gloo ranks sharing the GPU: 20 seeds),
#   2. the whole single-rank parity suite and the distributed suite with the injection on from the first call, per seed.
# A result that changes under injection is a missing stream edge; the failing assertion names the seed.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
rc=0
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_distributed.py -x -q -k "delay_injection" > gpurun_out/race_inproc.log 2>&1 || rc=1
echo "in-process seeds: rc=$rc"; tail -3 gpurun_out/race_inproc.log
for seed in ${SEEDS:-11 12}; do
  SMCP_RACE=$seed timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -p no:cacheprovider > gpurun_out/race_parity_$seed.log 2>&1 || rc=1
  echo "parity suite, SMCP_RACE=$seed: $(tail -1 gpurun_out/race_parity_$seed.log)"
  [ $rc -ne 0 ] && break
  SMCP_RACE=$seed timeout -k 10 900 python3 -m pytest tests/test_gpu_distributed.py -x -q -p no:cacheprovider -k "not bench_self_launch" > gpurun_out/race_dist_$seed.log 2>&1 || rc=1
  echo "distributed suite, SMCP_RACE=$seed: $(tail -1 gpurun_out/race_dist_$seed.log)"
  [ $rc -ne 0 ] && break
done
exit $rc

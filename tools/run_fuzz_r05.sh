# round-5 randomised sweeps on new seeds (one gpurun call): parity vs the oracle, sharded with both forms of the top's exchange, IPM runs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python3 tools/fuzz_long.py ${NPAR:-600} ${SEED:-180000} > gpurun_out/fuzz5_long.log 2>&1; echo "fuzz_long rc=$?"; tail -2 gpurun_out/fuzz5_long.log
timeout -k 10 300 python3 tools/fuzz_sharded.py 60 181000 2 > gpurun_out/fuzz5_sharded2.log 2>&1; echo "fuzz_sharded(2) rc=$?"; tail -1 gpurun_out/fuzz5_sharded2.log
SMCP_SHARD_TOP=constraint timeout -k 10 300 python3 tools/fuzz_sharded.py 60 182000 2 > gpurun_out/fuzz5_sharded2c.log 2>&1; echo "fuzz_sharded(2, top by constraint) rc=$?"; tail -1 gpurun_out/fuzz5_sharded2c.log
SMCP_SHARD_TOP=constraint timeout -k 10 300 python3 tools/fuzz_sharded.py 40 183000 3 > gpurun_out/fuzz5_sharded3c.log 2>&1; echo "fuzz_sharded(3, top by constraint) rc=$?"; tail -1 gpurun_out/fuzz5_sharded3c.log
timeout -k 10 300 python3 tools/fuzz_sharded.py 40 184000 4 > gpurun_out/fuzz5_sharded4.log 2>&1; echo "fuzz_sharded(4, default = top by constraint) rc=$?"; tail -1 gpurun_out/fuzz5_sharded4.log
timeout -k 10 900 python3 tools/fuzz_ipm.py 24 185000 > gpurun_out/fuzz5_ipm.log 2>&1; echo "fuzz_ipm rc=$?"; tail -1 gpurun_out/fuzz5_ipm.log
timeout -k 10 300 python3 tools/fuzz_maxcut.py > gpurun_out/fuzz5_maxcut.log 2>&1; echo "fuzz_maxcut rc=$?"; tail -2 gpurun_out/fuzz5_maxcut.log
timeout -k 10 120 python3 tools/ipm_synth50k.py > gpurun_out/ipm_synth50k_r05.log 2>&1; tail -2 gpurun_out/ipm_synth50k_r05.log
timeout -k 10 120 python3 tools/maxcut.py 1000 5909 > gpurun_out/maxcut_r05.log 2>&1; tail -2 gpurun_out/maxcut_r05.log

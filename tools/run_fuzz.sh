# randomised sweeps on the code as it stands (one gpurun call): parity, sharded, whole interior-point runs, end-to-end timings
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 420 python3 tools/fuzz_long.py ${NPAR:-400} ${SEED:-160000} > gpurun_out/fuzz_long_r04.log 2>&1; echo "fuzz_long rc=$?"; tail -2 gpurun_out/fuzz_long_r04.log
timeout -k 10 300 python3 tools/fuzz_sharded.py 60 161000 2 > gpurun_out/fuzz_sharded2_r04.log 2>&1; echo "fuzz_sharded(2) rc=$?"; tail -2 gpurun_out/fuzz_sharded2_r04.log
timeout -k 10 300 python3 tools/fuzz_sharded.py 40 162000 3 > gpurun_out/fuzz_sharded3_r04.log 2>&1; echo "fuzz_sharded(3) rc=$?"; tail -2 gpurun_out/fuzz_sharded3_r04.log
timeout -k 10 600 python3 tools/fuzz_ipm.py 24 163000 > gpurun_out/fuzz_ipm_r04.log 2>&1; echo "fuzz_ipm rc=$?"; tail -2 gpurun_out/fuzz_ipm_r04.log
timeout -k 10 120 python3 tools/ipm_synth50k.py > gpurun_out/ipm_synth50k_r04.log 2>&1; tail -2 gpurun_out/ipm_synth50k_r04.log
timeout -k 10 120 python3 tools/maxcut.py 1000 5909 > gpurun_out/maxcut_r04.log 2>&1; tail -2 gpurun_out/maxcut_r04.log

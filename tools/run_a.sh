# new-path tests, then bench (fused vs split calls), then a step timeline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_parity.py -x -q -k "dual_scaling or deferred_potrf or kkt_factor_and_solve or deferred_status" > gpurun_out/t_a.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/t_a.log
python3 bench.py --steps 20 --warmup 5 --no-cpu --no-secondary > gpurun_out/b_fused.json 2> gpurun_out/b_fused.err; echo "fused rc=$?"; cut -c1-400 gpurun_out/b_fused.json
SMCP_BENCH_SPLIT=1 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-secondary > gpurun_out/b_split.json 2> gpurun_out/b_split.err; echo "split rc=$?"; cut -c1-400 gpurun_out/b_split.json
bash tools/trace_step.sh > gpurun_out/trace_a.log 2>&1; echo "trace rc=$?"; tail -3 gpurun_out/trace_a.log

# GPU suite + default bench line + the two-partition emulation:  bash tools/run_checks.sh  (one gpurun call)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 -m pytest tests -m gpu -x -q > gpurun_out/gputests.log 2>&1; echo "gpu tests rc=$?" ; tail -3 gpurun_out/gputests.log
python3 bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/bench_default.err
python3 tools/shard_step_emul.py --mode both > gpurun_out/shard_emul.log 2>&1; echo "emul rc=$?"; tail -12 gpurun_out/shard_emul.log

# a second, larger randomised sweep on other seeds (one gpurun call): parity, sharded with two / three / four ranks, interior-point runs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 700 python3 tools/fuzz_long.py ${NPAR:-900} ${SEED:-170000} > gpurun_out/fuzz2_long.log 2>&1; echo "fuzz_long rc=$?"; tail -2 gpurun_out/fuzz2_long.log
timeout -k 10 300 python3 tools/fuzz_sharded.py 100 171000 2 > gpurun_out/fuzz2_sharded2.log 2>&1; echo "fuzz_sharded(2) rc=$?"; tail -1 gpurun_out/fuzz2_sharded2.log
timeout -k 10 300 python3 tools/fuzz_sharded.py 60 172000 3 > gpurun_out/fuzz2_sharded3.log 2>&1; echo "fuzz_sharded(3) rc=$?"; tail -1 gpurun_out/fuzz2_sharded3.log
timeout -k 10 300 python3 tools/fuzz_sharded.py 60 173000 4 > gpurun_out/fuzz2_sharded4.log 2>&1; echo "fuzz_sharded(4) rc=$?"; tail -1 gpurun_out/fuzz2_sharded4.log
timeout -k 10 900 python3 tools/fuzz_ipm.py 48 174000 > gpurun_out/fuzz2_ipm.log 2>&1; echo "fuzz_ipm rc=$?"; tail -1 gpurun_out/fuzz2_ipm.log
timeout -k 10 300 python3 tools/fuzz_maxcut.py > gpurun_out/fuzz2_maxcut.log 2>&1; echo "fuzz_maxcut rc=$?"; tail -2 gpurun_out/fuzz2_maxcut.log
python3 bench.py --workload maxcut --steps 10 --warmup 3 --no-cpu --no-secondary 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('maxcut', d['value'], d['roofline'])"

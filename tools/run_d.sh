cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_fuzz.py -x -q > gpurun_out/t_d.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|FAILED|smcp_amd:" gpurun_out/t_d.log | head -8
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-secondary --cpu-repeats 1 --verbose > gpurun_out/b_d.json 2> gpurun_out/b_d.err; echo "bench rc=$?"
python3 -c "
import json
d=json.loads(open('gpurun_out/b_d.json').read().strip().splitlines()[-1])
print(d['value'], d['value_as_allocated'], d['ms_per_step'], d['back_solve']['ms'], d['cpu_baseline']['gpu_vs_oracle_relerr'])
for k,v in list(d['kernel_ms_per_step'].items())[:14]: print('  %-28s %.4f'%(k,v))
"

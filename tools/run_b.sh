# quick A/B of environment variants on the headline step:  bash tools/run_b.sh "VAR1=x VAR2=y" "VAR3=z" ...   (TESTS=0: skip the tests)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ "${TESTS:-1}" = "1" ]; then
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_fuzz.py -x -q > gpurun_out/t_b.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|FAILED|smcp_amd:" gpurun_out/t_b.log | head -8
fi
i=0
for v in "" "$@"; do
  env $v timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-secondary --cpu-repeats 1 > gpurun_out/b_var$i.json 2> gpurun_out/b_var$i.err
  echo "[$v] rc=$? $(python3 -c "import json,sys; d=json.loads(open('gpurun_out/b_var$i.json').read().strip().splitlines()[-1]); print(d['value'], d['value_as_allocated'], d['ms_per_step'], d['back_solve']['ms'], d['cpu_baseline']['gpu_vs_oracle_relerr'], d['cpu_baseline'].get('schur_vs_oracle_relerr'), {k:v for k,v in list(d['kernel_ms_per_step'].items())[:5]})")"
  i=$((i+1))
done

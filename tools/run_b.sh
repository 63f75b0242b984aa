# quick A/B of environment variants on the headline step:  bash tools/run_b.sh "VAR1=x VAR2=y" "VAR3=z" ...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q > gpurun_out/t_b.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/t_b.log
i=0
for v in "" "$@"; do
  env $v python3 bench.py --steps 20 --warmup 5 --no-cpu --no-secondary > gpurun_out/b_var$i.json 2> gpurun_out/b_var$i.err
  echo "[$v] rc=$? $(python3 -c "import json,sys; d=json.loads(open('gpurun_out/b_var$i.json').read().strip().splitlines()[-1]); print(d['value'], d['value_as_allocated'], d['ms_per_step'], d['back_solve']['ms'], d['config']['placement_tuning'])")"
  i=$((i+1))
done
bash tools/trace_step.sh > gpurun_out/trace_b.log 2>&1; echo "trace rc=$?"
python3 bench.py --workload maxcut --steps 10 --warmup 2 --no-cpu --no-secondary > gpurun_out/b_maxcut.json 2> gpurun_out/b_maxcut.err; echo "maxcut rc=$?"; cut -c1-300 gpurun_out/b_maxcut.json

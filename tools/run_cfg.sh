# GPU suite + the headline and the secondary workloads one by one (GPU legs only):  bash tools/run_cfg.sh [workloads...]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ "${TESTS:-1}" = "1" ]; then
python3 -m pytest tests -m gpu -x -q > gpurun_out/gputests.log 2>&1; echo "gpu tests rc=$?"; tail -2 gpurun_out/gputests.log
fi
for w in ${@:-synth50k dense4096 arrow maxcut}; do
  python3 bench.py --workload $w --steps 10 --warmup 3 --no-cpu --no-secondary > gpurun_out/b_$w.json 2> gpurun_out/b_$w.err
  python3 -c "
import json; d=json.loads(open('gpurun_out/b_$w.json').read().strip().splitlines()[-1]); print('$w', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], [(k, round(v, 3)) for k, v in list(d['kernel_ms_per_step'].items())[:7]])"
done

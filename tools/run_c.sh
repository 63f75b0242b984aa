# tests on the new plan / potrf paths, the IPM run with a host profile
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_fuzz.py -x -q > gpurun_out/t_c.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/t_c.log
python3 bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/b_c.json 2> gpurun_out/b_c.err; echo "bench rc=$?"
python3 -c "
import json
d=json.loads(open('gpurun_out/b_c.json').read().strip().splitlines()[-1])
print(d['value'], d['value_as_allocated'], d['ms_per_step'], d['back_solve']['ms'])
for n,s in d['secondary'].items(): print(n, s.get('value'), s.get('ms_per_step'), s.get('error'))
"
python3 tools/ipm_synth50k.py > gpurun_out/ipm_plain.log 2>&1; tail -3 gpurun_out/ipm_plain.log
python3 -m cProfile -o gpurun_out/ipm.prof tools/ipm_synth50k.py > gpurun_out/ipm_prof.log 2>&1; tail -2 gpurun_out/ipm_prof.log
python3 -c "
import pstats
p=pstats.Stats('gpurun_out/ipm.prof'); p.sort_stats('tottime').print_stats(35)
" > gpurun_out/ipm_prof_top.txt 2>&1
python3 -c "
import pstats
p=pstats.Stats('gpurun_out/ipm.prof'); p.sort_stats('cumtime').print_stats(60)
" > gpurun_out/ipm_prof_cum.txt 2>&1

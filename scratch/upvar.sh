for sk in 0 128; do
SMCP_SKIP=$sk python bench.py --steps 5 --no-cpu > gpurun_out/b_$sk.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/b_$sk.json').read().strip().splitlines()[-1]); print('skip $sk', d['ms_per_step'], d['kernel_ms_per_step']['k_hess_up_pad'])"
done

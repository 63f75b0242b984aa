#!/bin/bash
# two ranks on ONE GPU is only a functional check of the N > 1 bench path (gloo would be needed for real sharing;
# RCCL refuses two ranks on one device), so run it with world=2 over gloo by env override
export SMCP_BENCH_BACKEND=gloo
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --workload synth6k --no-cpu "$@"

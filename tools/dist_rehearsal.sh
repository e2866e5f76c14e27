#!/bin/bash
# Two ranks on ONE GPU over gloo: rehearses bench.py's multi-rank control flow (not RCCL itself).
export BENCH_DIST_BACKEND=gloo PYTHONFAULTHANDLER=1 BENCH_TRACE=1
timeout -k 5 70 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
  --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 2 > gpurun_out/dist_rehearsal.log 2>&1
echo "exit $?" >> gpurun_out/dist_rehearsal.log

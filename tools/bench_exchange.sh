#!/bin/bash
# The N > 1 step of bench.py rehearsed on a one-rank RCCL communicator (BENCH_FORCE_DIST=1), once per
# way of exchanging the ids: started by the GPU behind a watcher kernel (the default), the library's
# communicator behind an event record + stream wait, torch.distributed, and no collective at all.
export MASTER_ADDR=127.0.0.1 MASTER_PORT=${MASTER_PORT:-29517} RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
CFG=${1:-metric}
for mode in device native torch; do
  echo "exchange=$mode"
  BENCH_FORCE_DIST=1 BENCH_EXCHANGE=$mode timeout -k 10 200 python3 bench.py --no-cpu-baseline --config "$CFG" | cut -c1-260 || exit 1
done
echo "no collective"
timeout -k 10 200 python3 bench.py --no-cpu-baseline --config "$CFG" | cut -c1-200

#!/bin/bash
# Which part of the exchange costs the compute stream time (one-rank RCCL communicator).
# MI355_DP_EXCHANGE_PROBE bit 1 drops the collective, 2 the completion event, 4 the side stream's wait
# (or the watcher kernel), 8 the event record on the compute stream and the wait.
export MASTER_ADDR=127.0.0.1 MASTER_PORT=${MASTER_PORT:-29519} RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for w in ${WATCHER:-0 1}; do
for probe in ${PROBES:-0 1 2 3 4 5 7 8 11}; do
  echo -n "watcher=$w probe=$probe  "
  BENCH_EXCHANGE=$([ $w = 1 ] && echo device || echo native) MI355_DP_EXCHANGE_PROBE=$probe BENCH_FORCE_DIST=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 400 | grep -o '"ms_per_step": [0-9.]*, "host_issue_ms_per_step": [0-9.]*' || exit 1
done
done
echo -n "no collective  "
timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 400 | grep -o '"ms_per_step": [0-9.]*, "host_issue_ms_per_step": [0-9.]*'

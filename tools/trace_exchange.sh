#!/bin/bash
# Kernel traces of bench.py's step with the GPU-started exchange (one-rank communicator) and without
# a collective: where in the step the extra microseconds sit (tools/trace_gaps.py on each trace).
export MASTER_ADDR=127.0.0.1 MASTER_PORT=${MASTER_PORT:-29521} RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3
mkdir -p "$OUT"
cd /tmp
for mode in device native none; do
  if [ $mode = none ]; then unset BENCH_FORCE_DIST; else export BENCH_FORCE_DIST=1 BENCH_EXCHANGE=$mode MI355_DP_EXCHANGE_PROBE=${PROBE:-0}; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_x_$mode" -o x -- \
    python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --steps 100 > "$OUT/trace_x_$mode.txt" 2> "$OUT/trace_x_$mode.err" || exit 1
  f=$(find "$OUT/trace_x_$mode" -name '*kernel_trace.csv' | head -1)
  echo "== $mode"
  python3 "$GRAFT_REPO_ROOT/tools/trace_step_parts.py" "$f"
done

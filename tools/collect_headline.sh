#!/bin/bash
# Headline measurements of one library build, on the GPU box from the repo root:
#   tools/collect_headline.sh <tag>      (writes gpurun_out/<tag>/...)
# bench lines (metric config and config 2), rocprofv3 kernel traces of the same commands, and the
# HBM-traffic counter passes (FETCH_SIZE and WRITE_SIZE in separate runs), digested on the spot
# together with the sha256 of the library that ran.  Copy into profiles/ with
#   tools/publish_headline.py <tag>
set -e
TAG=${1:-r2}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
LIB=vllm-from-scratch_amd/mi355_paged/libmi355_paged.so
SHA=$(sha256sum "$LIB" | cut -d' ' -f1)
echo "$SHA" > "$OUT/library_sha256.txt"
step() { echo "== $*" >&2; }

for cfg in metric c2; do
  step "counters $cfg"
  # (counter passes serialise the launches: a short clock warm-up; bytes per launch do not depend on clocks)
  export BENCH_CLOCK_WARM_STEPS=50
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_$cfg" -o fetch -- \
    python3 bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2> "$OUT/pmc_fetch_$cfg.log"
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_$cfg" -o write -- \
    python3 bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2> "$OUT/pmc_write_$cfg.log"
  python3 tools/pmc_traffic.py "$(find "$OUT/pmc_fetch_$cfg" -name '*counter_collection.csv' | head -1)" \
    "$(find "$OUT/pmc_write_$cfg" -name '*counter_collection.csv' | head -1)" "$OUT/pmc_traffic_$cfg.json" "$SHA" > /dev/null
  unset BENCH_CLOCK_WARM_STEPS
  step "kernel trace $cfg"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$cfg" -o $cfg -- \
    python3 bench.py --config $cfg --steps 100 --no-cpu-baseline > "$OUT/bench_line_${cfg}_under_rocprof.json" 2> "$OUT/trace_$cfg.log"
done
step bench lines
# the digests above sit where bench.py looks for them only after publishing; point it at them
mkdir -p profiles
cp "$OUT/pmc_traffic_metric.json" "profiles/${TAG}_pmc_traffic.json"
cp "$OUT/pmc_traffic_c2.json" "profiles/${TAG}_pmc_traffic_c2.json"
timeout -k 10 300 python3 bench.py > "$OUT/bench_line.json" 2> "$OUT/bench.err"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_line_steps20.json" 2>> "$OUT/bench.err"
timeout -k 10 300 python3 bench.py --config c2 --no-cpu-baseline > "$OUT/bench_line_c2.json" 2>> "$OUT/bench.err"
cat "$OUT/bench_line.json" "$OUT/bench_line_steps20.json" "$OUT/bench_line_c2.json"

#!/bin/bash
# Collect the measurements kept under profiles/: run on the GPU box from the repo root,
#   tools/collect_profiles.sh <tag> [traces|secondary|secondary1|secondary2|all]      (writes gpurun_out/<tag>/...)
# Kernel-trace passes and counter (--pmc) passes are separate rocprofv3 runs.  The two halves fit a
# 20-minute GPU call each: "traces" = headline set, kernel traces, counters; "secondary" = the rest (secondary1 + secondary2: secondary_a.txt, secondary_b.txt).
set -e
TAG=${1:-r1f}
PART=${2:-all}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
step() { echo "== $*" >&2; }

if [ "$PART" = traces ] || [ "$PART" = all ]; then
step headline: bench lines, kernel traces and HBM-traffic counters of the metric config and config 2
tools/collect_headline.sh "$TAG" > "$OUT/headline.txt" 2> "$OUT/headline.err"

step kernel trace of the small decode config, the grouped-query, MLA and prefill benches
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_c2tune" -o c2tune -- \
  python3 tools/tune_c2.py c2 > "$OUT/c2_under_rocprof.txt" 2> "$OUT/trace_c2tune.log"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_gqa" -o gqa -- \
  python3 tools/bench_gqa.py > "$OUT/gqa_under_rocprof.txt" 2> "$OUT/trace_gqa.log"
MLA_TARGETS=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_mla" -o mla -- \
  python3 tools/bench_mla.py 256 16 4096 > "$OUT/mla_under_rocprof.txt" 2> "$OUT/trace_mla.log"
MLA_TARGETS=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_mla_wide" -o mla_wide -- \
  python3 tools/bench_mla.py 128 128 4096 > "$OUT/mla_wide_under_rocprof.txt" 2> "$OUT/trace_mla_wide.log"
BENCH_MFMA_ONLY=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_prefill" -o prefill -- \
  python3 tools/bench_prefill.py 4 8192 32 8 128 > "$OUT/prefill_under_rocprof.txt" 2> "$OUT/trace_prefill.log"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_paged_prefill" -o pp -- \
  python3 tools/bench_paged_prefill.py > "$OUT/paged_prefill_under_rocprof.txt" 2> "$OUT/trace_paged_prefill.log"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_engine" -o engine -- \
  python3 tools/bench_engine_mixed.py 512 128 2048 > "$OUT/engine_under_rocprof.txt" 2> "$OUT/trace_engine.log"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_prefill_f32" -o prefill_f32 -- \
  python3 tools/bench_prefill_fp32.py 16 2048 8 8 64 > "$OUT/prefill_f32_under_rocprof.txt" 2> "$OUT/trace_prefill_f32.log"
step kernels of one captured decode step of the engine
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_engine_decode" -o engdec -- \
  python3 tools/bench_engine.py 256 512 32 > "$OUT/engine_decode_under_rocprof.txt" 2> "$OUT/trace_engine_decode.log"
python3 tools/trace_decode_step.py "$(find "$OUT/trace_engine_decode" -name '*kernel_trace.csv' | head -1)" > "$OUT/engine_decode_step_kernels.txt"
step "kernel trace of the one-rank rehearsal of the multi-rank step, exchange started by the GPU"
MASTER_ADDR=127.0.0.1 MASTER_PORT=29519 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 BENCH_FORCE_DIST=1 \
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_exchange" -o exchange -- \
  python3 bench.py --no-cpu-baseline --steps 100 > "$OUT/bench_line_exchange_under_rocprof.json" 2> "$OUT/trace_exchange.log"
if [ -f tools/_ab/lib_wide_timing.so ]; then
  step phases of a step of the 128-heads-per-workgroup MLA kernel from the stamped side build
  MI355_PAGED_LIB=$PWD/tools/_ab/lib_wide_timing.so timeout -k 10 200 python3 tools/probe_wide_timeline.py > "$OUT/mla_wide_phases.txt" 2> "$OUT/mla_wide_phases.err"
  echo "-- 128 rows: two token splits merged in the launch (split 0 shorter: it writes a record, split 1 combines)" >> "$OUT/mla_wide_phases.txt"
  MI355_PAGED_LIB=$PWD/tools/_ab/lib_wide_timing.so timeout -k 10 200 python3 tools/probe_wide_timeline.py 128 128 4096 >> "$OUT/mla_wide_phases.txt" 2>> "$OUT/mla_wide_phases.err"
fi
if [ -f tools/_ab/lib_samp_t.so ]; then
  step sampler phases from the stamped side build
  MI355_PAGED_LIB=tools/_ab/lib_samp_t.so timeout -k 10 200 python3 tools/probe_sampler_phases.py > "$OUT/sampler_phases.txt" 2> "$OUT/sampler_phases.err"
fi

step counters: MFMA activity of the prefill kernels, MLA
export MI355_WARM_SECONDS=0.05 MLA_WARM_SECONDS=0.05   # (counter passes serialise launches; counts do not depend on clocks)
BENCH_MFMA_ONLY=1 tools/pmc_sq.sh "$OUT/pmc_prefill_d128" prefill_mfma tools/bench_prefill.py 4 8192 32 8 128 > "$OUT/pmc_prefill_d128.txt"
BENCH_MFMA_ONLY=1 tools/pmc_sq.sh "$OUT/pmc_prefill_d64" prefill_mfma tools/bench_prefill.py 2 16384 8 8 64 > "$OUT/pmc_prefill_d64.txt"
tools/pmc_sq.sh "$OUT/pmc_prefill_f32" prefill_mfma_f32 tools/bench_prefill_fp32.py 16 2048 8 8 64 > "$OUT/pmc_prefill_f32.txt"
MLA_TARGETS=0 tools/pmc_sq.sh "$OUT/pmc_mla_wide" mla_decode_kernel_wide tools/bench_mla.py 256 128 4096 > "$OUT/pmc_mla_wide.txt"
MI355_MLA_WIDE=0 MLA_TARGETS=0 tools/pmc_sq.sh "$OUT/pmc_mla_mh" mla_decode_kernel_mh tools/bench_mla.py 128 128 4096 > "$OUT/pmc_mla_mh.txt"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_mla_fetch" -o fetch -- \
  python3 tools/bench_mla.py 256 16 4096 > /dev/null 2> "$OUT/pmc_mla_fetch.log"

# what comes back from the box is capped at 64 MiB: the per-launch traces have been digested above
find "$OUT" -name '*kernel_trace.csv' -size +1M -delete
fi
if [ "$PART" = traces ]; then exit 0; fi
unset MI355_WARM_SECONDS MLA_WARM_SECONDS
step secondary benchmarks
if [ "$PART" != secondary2 ]; then
{
  timeout -k 10 200 python3 tools/bench_misc.py
  timeout -k 10 200 python3 tools/bench_sampler.py
  timeout -k 10 200 python3 tools/bench_sampler.py 256 128256
  timeout -k 10 200 python3 tools/bench_gqa.py
  timeout -k 10 200 python3 tools/bench_locality.py
  timeout -k 10 200 python3 tools/tune_c2.py c2
  timeout -k 10 200 python3 tools/tune_c2.py metric
  timeout -k 10 200 python3 tools/bench_argmax.py
  timeout -k 10 200 python3 tools/bench_argmax.py 256 1000
  for a in "64 16 4096" "256 16 4096" "128 16 8192" "64 32 4096" "32 64 4096" "128 128 4096" "256 128 4096" "256 64 4096" "256 32 4096" "64 128 8192"; do MLA_TARGETS=0,0 timeout -k 10 120 python3 tools/bench_mla.py $a | tail -1; done
  echo "more than 16 heads on the 64-heads-per-workgroup kernel (MI355_MLA_WIDE=0) instead of the 128-heads-per-workgroup one"
  for a in "64 32 4096" "32 64 4096" "128 128 4096" "256 128 4096" "256 64 4096" "256 32 4096"; do MI355_MLA_WIDE=0 MLA_TARGETS=0,0 timeout -k 10 120 python3 tools/bench_mla.py $a | tail -1; done
  echo "the same rows in randomly placed blocks of 16 tokens, a pool each for c_kv and k_rope (mi355_mla_decode_paged); one pool of R + P wide rows; fp32"
  for a in "64 16 4096" "256 16 4096" "128 128 4096"; do MLA_PAGED=16 MLA_PAGED_SPLIT=1 MLA_TARGETS=0 timeout -k 10 120 python3 tools/bench_mla.py $a | tail -1; done
  MLA_PAGED=16 MLA_TARGETS=0 timeout -k 10 120 python3 tools/bench_mla.py 256 16 4096 | tail -1
  MLA_PAGED=16 MLA_PAGED_SPLIT=1 MLA_TARGETS=0 timeout -k 10 120 python3 tools/bench_mla.py 256 16 4096 float32 | tail -1
  echo "a timing loop on a cold card (no clock warm-up), two loops in a row"
  MLA_WARM_SECONDS=0 MLA_TARGETS=0,0 timeout -k 10 120 python3 tools/bench_mla.py 256 16 4096 | tail -2
  MI355_MLA_MANY_HEADS=0 MLA_TARGETS=0 timeout -k 10 120 python3 tools/bench_mla.py 128 128 4096 | tail -1 | sed 's/^/16 heads per workgroup: /'
  timeout -k 10 200 python3 tools/bench_small_batch.py
  timeout -k 10 200 python3 tools/ab_fused_merge.py
  timeout -k 10 200 python3 tools/bench_ragged.py
  timeout -k 10 200 python3 tools/ab_balance.py
  for a in "16 2048 8 8 64" "2 16384 8 8 64" "16 2048 32 8 128" "4 8192 32 8 128" "64 512 32 8 128"; do BENCH_MFMA_ONLY=1 timeout -k 10 120 python3 tools/bench_prefill.py $a; done
  timeout -k 10 120 python3 tools/bench_prefill_fp32.py 16 2048 8 8 64
  timeout -k 10 120 python3 tools/bench_prefill_fp32.py 16 2048 4 4 32
  timeout -k 10 120 python3 tools/bench_prefill_fp32.py 4 8192 8 8 64
  timeout -k 10 120 python3 tools/bench_prefill_fp32.py 16 2048 8 2 128
  timeout -k 10 120 python3 tools/bench_paged_prefill.py 8 1024 3072 8 8 64 float32
  MLA_TARGETS=0 timeout -k 10 120 python3 tools/bench_mla.py 256 16 4096 float32 | tail -1
  for a in "8 1024 3072 32 8 128" "8 1024 3072 8 8 64" "32 256 256 8 8 64" "64 64 1024 8 8 64"; do timeout -k 10 120 python3 tools/bench_paged_prefill.py $a; done
} > "$OUT/secondary_a.txt" 2> "$OUT/secondary_a.err"
fi
if [ "$PART" = secondary1 ]; then exit 0; fi
{
  timeout -k 10 200 python3 tools/bench_dtypes.py
  timeout -k 10 200 python3 tools/bench_engine.py 256 512 32
  echo "... with the decode branch as RoPE + cache write, then attention (two launches) instead of one"
  ENGINE_FUSED_DECODE=0 timeout -k 10 200 python3 tools/bench_engine.py 256 512 32
  timeout -k 10 200 python3 tools/bench_engine.py 128 256 64
  timeout -k 10 200 python3 tools/bench_engine.py 8 4096 8 1024
  timeout -k 10 200 python3 tools/bench_engine_mixed.py 512 128 2048
  timeout -k 10 200 python3 tools/time_mixed_steps.py 512 128 2048
  timeout -k 10 200 python3 tools/bench_reference_engine_config.py
  export MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
  echo "one-rank RCCL communicator (BENCH_FORCE_DIST=1), one collective per step: exchange started by the GPU (default), library communicator behind event + wait, torch.distributed; then no collective; twice"
  tools/bench_exchange.sh metric || echo "exchange rehearsal failed"
  tools/bench_exchange.sh metric || echo "exchange rehearsal failed"
  echo "... one collective per 4 steps (library communicator)"
  BENCH_FORCE_DIST=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --gather-every 4 | cut -c1-260
  echo "cache write + attention in one launch (--fused-step) against the two launches of the default step: metric config, config 2"
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --fused-step
  timeout -k 10 200 python3 bench.py --no-cpu-baseline
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --config c2 --fused-step
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --config c2
  echo "fixed total batch (--strong 2048) on one GPU: 2048 sequences in one launch"
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --strong 2048
} > "$OUT/secondary_b.txt" 2> "$OUT/secondary_b.err"
find "$OUT" -name '*.csv' | head -40 >&2

#!/bin/bash
# The 128-heads-per-workgroup MLA kernel against the 64-heads-per-workgroup one below 65 heads.
mkdir -p gpurun_out/r3w
for shape in "32 64 4096" "64 64 4096" "256 64 4096" "64 32 4096" "256 32 4096" "64 48 4096"; do
  set -- $shape
  for least in 65 17; do
    MI355_MLA_WIDE=$least MLA_TARGETS=0 timeout -k 10 100 python tools/bench_mla.py $shape 2>&1 | grep mla_decode | sed "s|^|wide from $least heads: |" || exit 1
  done
done | tee gpurun_out/r3w/ab_heads.log

#!/bin/bash
# Small batches at 128 heads: the 64-heads-per-workgroup MLA kernel (its own plan) against the
# 128-heads-per-workgroup one forced (MI355_MLA_WIDE_SPLIT_TOKENS=1) at several workgroup targets.
mkdir -p gpurun_out/r3w
for shape in "1 128 4096" "1 128 32768" "4 128 4096" "4 128 32768" "16 128 4096" "16 128 8192" "32 128 4096" "64 128 4096" "8 32 8192" "8 64 8192"; do
  MI355_MLA_WIDE=0 MLA_TARGETS=0 timeout -k 10 100 python tools/bench_mla.py $shape 2>&1 | grep mla_decode | sed "s|^|64-head kernel: |" || exit 1
  MI355_MLA_WIDE_SPLIT_TOKENS=1 MLA_TARGETS=16,32,64,128,256 timeout -k 10 100 python tools/bench_mla.py $shape 2>&1 | grep mla_decode | sed "s|^|128-head kernel: |" || exit 1
done | tee gpurun_out/r3w/ab_small.log

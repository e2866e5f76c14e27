#!/bin/bash
# Batches of 128 heads that need token splits to fill the chip: the 64-heads-per-workgroup MLA kernel
# (MI355_MLA_WIDE=0) against the library's plan (the 128-head kernel from 512 tokens per workgroup on).
mkdir -p gpurun_out/r3w
for shape in "16 128 4096" "24 128 4096" "32 128 4096" "40 128 4096" "48 128 4096" "64 128 4096" "16 128 8192" "24 128 8192" "8 128 16384" "4 128 32768" "1 128 32768"; do
  MI355_MLA_WIDE=0 MLA_TARGETS=0 timeout -k 10 100 python tools/bench_mla.py $shape 2>&1 | grep mla_decode | sed "s|^|64-head kernel: |; s/R=512 P=64//; s/GB\/s.*TB\/s)//; s/mla_decode bfloat16//" || exit 1
  MLA_TARGETS=0 timeout -k 10 100 python tools/bench_mla.py $shape 2>&1 | grep mla_decode | sed "s|^|planned:        |; s/R=512 P=64//; s/GB\/s.*TB\/s)//; s/mla_decode bfloat16//" || exit 1
done | tee gpurun_out/r3w/ab_small.log

#!/bin/bash
# The MLA plan for 128 heads over row counts around the chip's 256 CUs, for each library given
# (side builds under tools/_ab/; none: the tree's), alternating.
mkdir -p gpurun_out/r3w
libs=${@:-vllm-from-scratch_amd/mi355_paged/libmi355_paged.so}
for shape in "128 128 4096" "96 128 4096" "64 128 8192" "100 128 4096" "160 128 4096" "200 128 4096" "48 128 4096" "64 128 4096" "128 64 4096" "128 32 4096" "32 128 16384"; do
  for lib in $libs; do
    MI355_PAGED_LIB=$PWD/$lib MLA_TARGETS=0 timeout -k 10 100 python tools/bench_mla.py $shape 2>&1 |
      grep mla_decode | sed "s|^|$(basename $lib): |; s/R=512 P=64//; s/GB\/s.*TB\/s)//; s/mla_decode bfloat16//" || exit 1
  done
done | tee gpurun_out/r3w/ab_plan.log

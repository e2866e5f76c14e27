#!/bin/bash
# The 128-heads-per-workgroup MLA kernel with two token splits at 128 heads x 128 / 96 rows x 4096
# tokens: the merge kernel (MI355_MLA_WIDE_FUSED_MERGE=0) against the merge inside the launch with
# split 0 shortened by 0 .. 160 tokens, for each library given (side builds under tools/_ab/; none:
# the tree's), alternating.
mkdir -p gpurun_out/r3w
libs=${@:-vllm-from-scratch_amd/mi355_paged/libmi355_paged.so}
for rep in 1 2; do
  for lib in $libs; do
    for mode in "0 0" "1 0" "1 32" "1 64" "1 96" "1 160"; do
      set -- $mode
      for shape in "128 128 4096" "96 128 4096"; do
        MI355_PAGED_LIB=$PWD/$lib MI355_MLA_WIDE_FUSED_MERGE=$1 MI355_MLA_WIDE_SKEW_TOKENS=$2 MLA_TARGETS=0 timeout -k 10 100 python tools/bench_mla.py $shape 2>&1 |
          grep mla_decode | sed "s|^|$(basename $lib) in-launch=$1 skew=$2: |; s/R=512 P=64//; s/GB\/s.*TB\/s)//; s/mla_decode bfloat16//" || exit 1
      done
    done
  done
done | tee gpurun_out/r3w/ab_merge.log

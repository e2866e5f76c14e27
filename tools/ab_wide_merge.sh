#!/bin/bash
# The 128-heads-per-workgroup MLA kernel with two token splits at 128 heads x 128 rows x 4096 tokens:
# the merge kernel (MI355_MLA_WIDE_FUSED_MERGE=0) against the merge inside the launch over the tokens
# split 0 is shortened by (SHAPE="rows heads tokens" for another shape).
mkdir -p gpurun_out/r3w
for rep in 1 2 3; do
  for mode in "0 0" "1 0" "1 32" "1 64" "1 96" "1 160"; do
    set -- $mode
    MI355_MLA_WIDE_FUSED_MERGE=$1 MI355_MLA_WIDE_SKEW_TOKENS=$2 MLA_TARGETS=0 timeout -k 10 100 python tools/bench_mla.py ${SHAPE:-128 128 4096} 2>&1 |
      grep mla_decode | sed "s|^|in-launch=$1 skew=$2: |; s/R=512 P=64//; s/GB\/s.*TB\/s)//; s/mla_decode bfloat16//" || exit 1
  done
done | tee gpurun_out/r3w/ab_merge.log

#!/bin/bash
# Times the many-heads MLA shape on the tree's library and on each tools/_ab/lib_wide_*.so.
#   bash tools/ab_wide.sh [rows heads ctx]
shape=${@:-256 128 4096}
mkdir -p gpurun_out/r3w
for lib in "" tools/_ab/lib_wide_*.so; do
  MI355_PAGED_LIB=${lib:+$PWD/$lib} MLA_TARGETS=0 timeout -k 10 100 python tools/bench_mla.py $shape 2>&1 | grep mla_decode | sed "s|^|$(basename ${lib:-tree}): |" || exit 1
done | tee gpurun_out/r3w/ab.log

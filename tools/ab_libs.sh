#!/bin/bash
# Time the prompt-attention kernel of several builds of the library on one shape.
#   tools/ab_libs.sh "16 2048 8 8 64" lib1.so lib2.so ...   (the library in the tree is always first)
shape=$1; shift
for lib in "" "$@"; do
  MI355_PAGED_LIB=$lib BENCH_MFMA_ONLY=1 timeout -k 10 100 python tools/bench_prefill.py $shape 2>&1 | grep mfma | sed "s|^|$(basename ${lib:-tree}): |" || exit 1
done

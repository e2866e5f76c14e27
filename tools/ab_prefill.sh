#!/bin/bash
# A/B of the prompt-attention kernel: the library in the tree against another build of it.
#   tools/ab_prefill.sh path/to/other.so
set -e
for shape in "16 2048 8 8 64" "16 2048 32 8 128" "4 8192 32 8 128" "2 16384 8 8 64"; do
  for lib in "" "$1"; do
    MI355_PAGED_LIB=$lib timeout -k 10 100 python tools/bench_prefill.py $shape 2>&1 | grep mfma | sed "s|^|${lib:-tree}: |"
  done
done

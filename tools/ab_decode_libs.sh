#!/bin/bash
# A/B of builds of the library on the decode launch (tools/tune_c2.py: config 2 and the metric
# shape), alternating processes, two rounds.
#   tools/ab_decode_libs.sh lib1.so lib2.so ...
set -e
for rep in 1 2; do
  for lib in "$@"; do
    for cfg in c2 metric; do
      MI355_PAGED_LIB=$lib timeout -k 10 120 python tools/tune_c2.py $cfg 2>&1 | grep "^$cfg" | sed "s|^|$(basename $lib): |"
    done
  done
done

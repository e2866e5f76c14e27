#!/bin/bash
# A/B of builds of the library on the decode launch (tools/tune_c2.py), alternating processes.
# AB_LIBS="a.so b.so" names the builds (default: tools/_ab/lib_base.so, a copy of an earlier build, and the in-tree one).
set -e
for rep in 1 2; do
  for lib in ${AB_LIBS:-tools/_ab/lib_base.so vllm-from-scratch_amd/mi355_paged/libmi355_paged.so}; do
    for cfg in c2 metric; do
      echo "== $lib $cfg"
      MI355_PAGED_LIB=$lib timeout -k 10 120 python tools/tune_c2.py $cfg
    done
  done
done

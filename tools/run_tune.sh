#!/bin/bash
# Sweep decode-kernel build variants x launch-plan targets on one GPU.
set -e
V=vllm-from-scratch_amd/mi355_paged/variants
for lib in "" $V/lib_u2nt1.so $V/lib_u8nt1.so $V/lib_u4nt0.so $V/lib_u2nt0.so; do
  export MI355_PAGED_LIB=$lib
  timeout -k 10 120 python tools/tune_decode.py metric 256 512 1024 2048
  timeout -k 10 120 python tools/tune_decode.py c2 64 256 512 1024 2048
done

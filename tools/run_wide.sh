#!/bin/bash
# Tests of the 128-heads-per-workgroup MLA kernel, then its timing beside the 64-head kernel's.
# Usage (on the GPU box, from the repository root): bash tools/run_wide.sh <tag> [test file]
tag=${1:-w}
tests=${2:-"test_gpu_mla.py -k wide"}
out=gpurun_out/r3w
mkdir -p $out
(cd tests && timeout -k 10 400 python -u -m pytest $tests -x -q --timeout 120 --timeout-method thread) > $out/t_$tag.log 2>&1
rc=$?
tail -15 $out/t_$tag.log
echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
{
  timeout -k 10 120 python tools/bench_mla.py 128 128 4096 &&
  MI355_MLA_WIDE=0 MLA_TARGETS=0 timeout -k 10 120 python tools/bench_mla.py 128 128 4096 &&
  MLA_TARGETS=0 timeout -k 10 120 python tools/bench_mla.py 256 128 4096 &&
  MLA_TARGETS=0 timeout -k 10 120 python tools/bench_mla.py 64 128 8192 &&
  MLA_TARGETS=0 MLA_PAGED=16 timeout -k 10 120 python tools/bench_mla.py 128 128 4096
} > $out/b_$tag.log 2>&1
rc=$?
cat $out/b_$tag.log
exit $rc

#!/bin/bash
# kernel time of the sampler's temperature-only mode for each side library given (kernel trace, min / avg)
set -e
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in "$@"; do
  export MI355_PAGED_LIB=$R/vllm-from-scratch_amd/mi355_paged/$lib
  export SAMPLER_MODES=temperature
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ko_$lib -o t -- python3 $R/tools/bench_sampler.py > $R/gpurun_out/ko_$lib.log 2>&1
  grep sample_kernel $R/gpurun_out/ko_$lib/t_kernel_stats.csv | awk -F'","' -v l=$lib '{print l, "calls", $2, "avg", $4, "min", $6}'
done

#!/bin/bash
# Phase times (tools/probe_wide_timeline.py) of every -DWIDE_TIMING=1 side build tools/_ab/lib_wide_t*.so
mkdir -p gpurun_out/r3w
for lib in tools/_ab/lib_wide_t*.so; do
  MI355_PAGED_LIB=$PWD/$lib timeout -k 10 100 python tools/probe_wide_timeline.py "$@" 2>&1 | grep steps | sed "s|^|$(basename $lib): |" || exit 1
done | tee gpurun_out/r3w/timeline.log

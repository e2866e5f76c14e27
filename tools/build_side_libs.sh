#!/bin/bash
# Side builds of the library that the phase probes read (run here, on the CPU, before a collection;
# the results go to tools/_ab/, which git ignores and gpurun ships):
#   lib_samp_t.so   -DSAMPLER_TIMING=1: sample_kernel stamps the shader clock at its phase boundaries
#                   (tools/probe_sampler_phases.py)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_ab /tmp/mi355_side_samp
make -C vllm-from-scratch_amd/csrc -j8 OBJDIR=/tmp/mi355_side_samp TARGET=../../tools/_ab/lib_samp_t.so EXTRA=-DSAMPLER_TIMING=1 > /tmp/mi355_side_samp/build.log 2>&1
ls -la tools/_ab/lib_samp_t.so
#   lib_wide_timing.so  -DWIDE_TIMING=1: mla_decode_kernel_wide stamps the phases of its steps (tools/probe_wide_timeline.py)
bash tools/build_wide_variants.sh timing "-DWIDE_TIMING=1"

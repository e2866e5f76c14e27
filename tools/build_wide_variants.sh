#!/bin/bash
# Side builds of the library that differ only in mla_decode_wide.hip's compile-time switches
# (run here, on the CPU): tools/build_wide_variants.sh name1 "-DFLAGS1" name2 "-DFLAGS2" ...
# -> tools/_ab/lib_wide_<name>.so, linked with the tree's other objects.
set -e
cd "$(dirname "$0")/../vllm-from-scratch_amd/csrc"
make -j8 > /dev/null
mkdir -p ../../tools/_ab /tmp/mi355_wide
others=$(ls build/*.o | grep -v "mla_decode_wide.o\|-hip-\|-host-")
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-kernarg-preload-count=16 $flags -save-temps=obj -c mla_decode_wide.hip -o /tmp/mi355_wide/wide_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_ab/lib_wide_$name.so $others /tmp/mi355_wide/wide_$name.o -Wl,-rpath,/opt/rocm/lib
  grep "vgpr_spill_count" /tmp/mi355_wide/mla_decode_wide-hip-amdgcn-amd-amdhsa-gfx950.s | sort | uniq -c | tr "\n" " " | sed "s/^/$name: /"; echo
done

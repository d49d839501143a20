#!/bin/bash
# tools/ab_build.sh NAME "-DFOO=1 ..."  — an alternative build of libmetamaps_hip.so with extra compiler flags into _ab/NAME/ (git-ignored; travels with gpurun).
# bench.py / the tests load it with MM_LIB_PATH=_ab/NAME/libmetamaps_hip.so: two kernel variants measured in turns on ONE box (tools/ab.sh).
set -e
cd "$(dirname "$0")/../metamaps_amd/csrc"
name=$1; flags=$2
out=../../_ab/$name; mkdir -p $out/_build
for f in mm_seq mm_index mm_map mm_post mm_synth mm_api; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-result $flags -c $f.hip -o $out/_build/$f.o ) &
done
wait
g++ -O3 -std=c++17 -fPIC -c host_pack.cpp -o $out/_build/host_pack.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libmetamaps_hip.so $out/_build/*.o -L/opt/rocm/lib -lrccl -lpthread -Wl,-rpath,/opt/rocm/lib
ls -la $out/libmetamaps_hip.so

#!/bin/bash
# Host code of the device library under AddressSanitizer.  `build` (here, no GPU needed) compiles the library with the host side
# instrumented (the device code is left alone: GPU sanitizers are not available) and tools/sanitize/device_paths.cpp against it;
# `run` (on the GPU box) runs that driver over the scenes.  Leak detection is off: the HIP runtime keeps its own allocations.
set -e
cd "$(dirname "$0")/.."
OUT=build_ab/asan
if [ "$1" = "build" ]; then
  mkdir -p $OUT
  SAN="-fsanitize=address -fno-gpu-sanitize -fno-omit-frame-pointer -g"
  C=cuda_raytracing_amd/csrc
  hipcc --offload-arch=gfx950 -O1 -std=c++17 -fPIC -shared -ffp-contract=off -mllvm -disable-machine-licm -fno-gpu-rdc -Wno-unused-function $SAN -Iinclude -I$C \
    $C/rt_device.hip $C/host_scene.cpp $C/rt_error.cpp -o $OUT/librt_amd.so
  hipcc -O1 -std=c++17 $SAN -Iinclude tools/sanitize/device_paths.cpp -L$OUT -lrt_amd '-Wl,-rpath,$ORIGIN' -o $OUT/device_paths
elif [ "$1" = "run" ]; then
  export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:exitcode=23
  timeout -k 10 300 $OUT/device_paths assets scenes/t_icosa.txt scenes/t_boxgrid.txt scenes/t_bezier.txt scenes/c3_orange.txt scenes/c4_lucy.txt scenes/c5_final.txt
else
  echo "usage: $0 build|run"; exit 2
fi

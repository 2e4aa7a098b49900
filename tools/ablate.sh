#!/bin/bash
# Build timing-only variants of the library (RT_ABLATE_* drop one part of the shadow test; results are
# wrong, only the time matters) into build_ab/.  Run them on the GPU box with RT_AMD_LIB=build_ab/<v>.so.
set -e
cd "$(dirname "$0")/.."
mkdir -p build_ab
B="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -shared -ffp-contract=off -mllvm -disable-machine-licm -fno-gpu-rdc -Iinclude -Icuda_raytracing_amd/csrc cuda_raytracing_amd/csrc/rt_device.hip cuda_raytracing_amd/csrc/host_scene.cpp cuda_raytracing_amd/csrc/rt_error.cpp"
$B -o build_ab/a0.so & $B -DRT_ABLATE_MESH -o build_ab/a1.so & $B -DRT_ABLATE_CYL -o build_ab/a2.so & $B -DRT_ABLATE_SPHERE -o build_ab/a3.so & wait
$B -DRT_ABLATE_RNG -o build_ab/a4.so & $B -DRT_ABLATE_ALL_SHADOW -o build_ab/a5.so & $B -DRT_UTIL_COUNTERS -o build_ab/util.so & wait

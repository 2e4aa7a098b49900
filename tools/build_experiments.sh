#!/bin/bash
# Builds the measurement variants of the library into build_ab/ (never shipped; run them on the GPU box with
# RT_AMD_LIB=build_ab/<name>.so, e.g. through tools/ab.sh):
#   exp.so    -DRT_EXPERIMENTS                       the product kernels + the three-kernel render (RT_AMD_DEFER=1)
#   util.so   -DRT_EXPERIMENTS -DRT_UTIL_COUNTERS    + section timers and lane-use probes in the render kernel ([time] / [util] lines on stderr)
#   steps.so  ... -DRT_PROBE_STEPS                   + per-step probes of the tree walks instead of the classes of traced shadow rays
# Extra compiler flags for all three: EXTRA="-D..." tools/build_experiments.sh
set -e
cd "$(dirname "$0")/.."
mkdir -p build_ab
B="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -shared -ffp-contract=off -mllvm -disable-machine-licm -fno-gpu-rdc -Iinclude -Icuda_raytracing_amd/csrc $EXTRA cuda_raytracing_amd/csrc/rt_device.hip cuda_raytracing_amd/csrc/host_scene.cpp cuda_raytracing_amd/csrc/rt_error.cpp"
id() { python3 __graft_entry__.py --build-id $EXTRA "$@"; }  # every variant carries the id of its own flags (rt_build_id)
$B -DRT_EXPERIMENTS -DRT_BUILD_ID="\"$(id -DRT_EXPERIMENTS)\"" -o build_ab/exp.so &
$B -DRT_EXPERIMENTS -DRT_UTIL_COUNTERS -DRT_BUILD_ID="\"$(id -DRT_EXPERIMENTS -DRT_UTIL_COUNTERS)\"" -o build_ab/util.so &
$B -DRT_EXPERIMENTS -DRT_UTIL_COUNTERS -DRT_PROBE_STEPS -DRT_BUILD_ID="\"$(id -DRT_EXPERIMENTS -DRT_UTIL_COUNTERS -DRT_PROBE_STEPS)\"" -o build_ab/steps.so &
wait
ls -la build_ab/exp.so build_ab/util.so build_ab/steps.so

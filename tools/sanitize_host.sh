#!/bin/bash
# Host-side code under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only; GPU sanitizers are not available):
# the scene loader on every scene of this repository (and the reference's shipped scenes when that tree is present), and the
# CPU oracle on small renders of them in three semantics presets.  Any report fails the script.
set -e
cd "$(dirname "$0")/.."
OUT=${TMPDIR:-/tmp}/rt_sanitize
mkdir -p $OUT
SAN="-O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer"
HOST="cuda_raytracing_amd/csrc/host_scene.cpp cuda_raytracing_amd/csrc/rt_error.cpp"
g++ -std=c++17 $SAN -Iinclude -Icuda_raytracing_amd/csrc tools/sanitize/load_scenes.cpp $HOST -o $OUT/load_scenes
gcc -std=c11 $SAN -ffp-contract=off -fopenmp -Iinclude -c oracle/cpu_ref.c -o $OUT/cpu_ref.o
g++ -std=c++17 $SAN -fopenmp -Iinclude -Icuda_raytracing_amd/csrc tools/sanitize/oracle_render.cpp $OUT/cpu_ref.o $HOST -o $OUT/oracle_render -lm
$OUT/load_scenes assets scenes/*.txt
REF=/root/reference/CUDA_RayTracing/resources
[ -d $REF ] && $OUT/load_scenes $REF $REF/scene*.txt
$OUT/oracle_render assets scenes/t_*.txt scenes/c1*.txt scenes/c3*.txt scenes/c4*.txt
echo "sanitizers: clean"

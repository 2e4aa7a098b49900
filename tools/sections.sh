#!/bin/bash
# Cycle share per code section and lane use of the render kernel, from the instrumented library (build_ab/util.so or steps.so,
# tools/build_experiments.sh), on the GPU box.  Usage: tools/sections.sh <lib.so> "<config> [bench flags]" ...
# Output: one block per configuration in gpurun_out/sections_<TAG>.log (TAG in the environment, default "x").
LIB=$1; shift
LOG=gpurun_out/sections_${TAG:-x}.log
mkdir -p gpurun_out
for c in "$@"; do
  echo "== $LIB $c" >> $LOG
  RT_AMD_LIB=$LIB timeout -k 10 200 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-strict 2>&1 | python -c "
import sys, json
seen = set()
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); print('ms_per_step', round(j['ms_per_step'], 2), 'kernel_ms', round(j['roofline']['kernel_ms_avg'], 2))
    elif ('[util]' in line or '[time]' in line) and line not in seen:
        seen.add(line); print(line.rstrip())
" >> $LOG || exit 1
done
cat $LOG

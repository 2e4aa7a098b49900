#!/bin/bash
# Time every library variant in build_ab/ on the headline frame (run on the GPU box).
# Full output goes to gpurun_out/ab.log; the timings are repeated in one block at the end.
mkdir -p gpurun_out
: > gpurun_out/ab.log
for v in "$@"; do
  echo "== $v" >> gpurun_out/ab.log
  RT_AMD_LIB=build_ab/$v.so timeout -k 10 120 python bench.py --scene ${SCENE:-c4_lucy} --steps 3 --warmup 1 --no-cpu-baseline --no-strict 2>&1 | python -c "
import sys, json
seen = set()
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); print('ms_per_step', round(j['ms_per_step'], 2), 'kernel_ms', round(j['roofline']['kernel_ms_avg'], 2))
    elif ('[util]' in line or '[time]' in line) and line not in seen:
        seen.add(line); print(line.rstrip())
" >> gpurun_out/ab.log || exit 1
done
grep -E "^==|ms_per_step" gpurun_out/ab.log | paste - -

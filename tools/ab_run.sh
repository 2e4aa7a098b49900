#!/bin/bash
# Time every library variant in build_ab/ on the headline frame (run on the GPU box).
for v in "$@"; do
  echo "== $v"
  RT_AMD_LIB=build_ab/$v.so timeout -k 10 120 python bench.py --scene ${SCENE:-c4_lucy} --steps 3 --warmup 1 --no-cpu-baseline --no-strict 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); print('ms_per_step', round(j['ms_per_step'], 2))
    elif '[util]' in line or '[time]' in line: print(line.rstrip())
" || exit 1
done

#!/bin/bash
# A/B timing of library builds on the GPU box: tools/ab.sh "<config> [bench flags]" lib1.so lib2.so ...   ("-" = the in-tree library)
# One line per library and configuration in gpurun_out/ab_<tag>.log (TAG in the environment, default "x").
CFG="$1"; shift
LOG=gpurun_out/ab_${TAG:-x}.log
mkdir -p gpurun_out
for v in "$@"; do
  if [ "$v" = "-" ]; then unset RT_AMD_LIB; else export RT_AMD_LIB=$v; fi
  timeout -k 10 200 python bench.py --config $CFG --steps ${STEPS:-4} --warmup 1 --no-cpu-baseline --no-strict 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); print('$v', '$CFG', 'ms_per_step', round(j['ms_per_step'], 2), 'kernel_ms', round(j['roofline']['kernel_ms_avg'], 2), 'traced_Mrays', round(j['rays_traced_per_frame'] / 1e6, 1))
" >> $LOG || exit 1
done

#!/bin/bash
# One-kernel render against the three-kernel render (RT_AMD_DEFER=1) on bench configurations (GPU box).
# Usage: tools/defer_ab.sh "headline" "c5 --spp 32" ...
for c in "$@"; do for d in 0 1; do RT_AMD_DEFER=$d timeout -k 10 250 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-strict 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); print('$c defer=$d ms_per_step', round(j['ms_per_step'], 2), 'kernel_ms', round(j['roofline']['kernel_ms_avg'], 2), 'rays', int(j['rays_resolved_per_frame']), int(j['rays_traced_per_frame']))
" || exit 1; done; done

#!/bin/bash
# HBM write / fetch bytes of the render kernel for one frame of a bench configuration (GPU box).
# Usage: tools/pmc_write.sh <tag> <lib.so|-> "<config> [bench flags]"      -> gpurun_out/pmcw_<tag>.txt
TAG=$1; LIB=$2; CFG=$3
OUT=gpurun_out/pmcw_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
[ "$LIB" != "-" ] && export RT_AMD_LIB=$LIB
B="python3 bench.py --config $CFG --no-cpu-baseline --no-strict --steps 1 --warmup 0"
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/write -- $B > $OUT/write.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $B > $OUT/fetch.log 2>&1 || exit 1
python3 - $OUT "$TAG $LIB $CFG" <<'PY' | tee gpurun_out/pmcw_$TAG.txt
import csv, glob, sys
out = {}
for f in glob.glob(sys.argv[1] + '/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'render_kernel' in r['Kernel_Name']:
            out[r['Counter_Name']] = out.get(r['Counter_Name'], 0) + float(r['Counter_Value'])
# FETCH_SIZE / WRITE_SIZE are reported in kilobytes
print(sys.argv[2], ' '.join(f"{k}={v:.4g}" for k, v in sorted(out.items())), f"write_GB={out.get('WRITE_SIZE', 0) * 1024 / 1e9:.2f} fetch_GB={out.get('FETCH_SIZE', 0) * 1024 / 1e9:.2f}")
PY

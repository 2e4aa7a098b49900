#!/bin/bash
# Instruction-fetch counters of one frame of a bench configuration, on the GPU box (one group per process).
# Usage: tools/pmc_ifetch.sh <tag> "<config> [bench flags]" [lib.so]     -> gpurun_out/pmcif_<tag>.json
TAG=$1; CFG="$2"
[ -n "$3" ] && export RT_AMD_LIB=$3
OUT=gpurun_out/pmcif_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="python3 bench.py --config $CFG --no-cpu-baseline --no-strict --steps 1 --warmup 0"
timeout -s KILL 150 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --output-format csv -d $OUT/a -- $B > $OUT/a.log 2>&1 || echo "pass a failed"
timeout -s KILL 150 rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/b -- $B > $OUT/b.log 2>&1 || echo "pass b failed"
timeout -s KILL 150 rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES --output-format csv -d $OUT/c -- $B > $OUT/c.log 2>&1 || echo "pass c failed"
python3 - $OUT <<'PY'
import csv, glob, json, os, sys
out = {}
for f in glob.glob(sys.argv[1] + '/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'render_kernel' in r['Kernel_Name']:
            out[r['Counter_Name']] = out.get(r['Counter_Name'], 0) + float(r['Counter_Value'])
json.dump(out, open(sys.argv[1] + '.json', 'w'), indent=1, sort_keys=True)
print(json.dumps(out, sort_keys=True))
PY
tail -3 $OUT/a.log

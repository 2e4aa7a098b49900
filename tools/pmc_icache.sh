#!/bin/bash
# Instruction-fetch counters of one frame of the headline bench (GPU box).  Usage: tools/pmc_icache.sh <tag> [lib]
set -e
TAG=${1:-q}
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
[ -n "$2" ] && export RT_AMD_LIB=$2
B="python3 bench.py --no-cpu-baseline --no-strict --steps 1 --warmup 0"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_IFETCH SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/sq -- $B > $OUT/sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES --output-format csv -d $OUT/sqc -- $B > $OUT/sqc.log 2>&1 || echo "sqc pass failed"
python3 - $OUT <<'PY'
import csv,glob,sys,json
out={}
for f in glob.glob(sys.argv[1]+'/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'render_kernel' in r['Kernel_Name']:
            out[r['Counter_Name']]=out.get(r['Counter_Name'],0)+float(r['Counter_Value'])
json.dump(out,open(sys.argv[1]+'/summary.json','w'),indent=1)
for k in sorted(out): print(k,out[k])
PY
tail -3 $OUT/sqc.log

#!/bin/bash
# Quick PMC snapshot of one frame of the headline bench (GPU box).  Usage: tools/pmc_quick.sh <tag> [lib]
set -e
TAG=${1:-q}
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
[ -n "$2" ] && export RT_AMD_LIB=$2
B="python3 bench.py --no-cpu-baseline --no-strict --steps 1 --warmup 0"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq -- $B > $OUT/sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/sq2 -- $B > $OUT/sq2.log 2>&1 || echo sq2 failed
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $B > $OUT/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/write -- $B > $OUT/write.log 2>&1
python3 - $OUT <<'PY'
import csv,glob,sys,collections,json
out={}
for f in glob.glob(sys.argv[1]+'/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'render_kernel' in r['Kernel_Name']:
            out[r['Counter_Name']]=out.get(r['Counter_Name'],0)+float(r['Counter_Value'])
json.dump(out,open(sys.argv[1]+'/summary.json','w'),indent=1)
for k in sorted(out): print(k,out[k])
PY

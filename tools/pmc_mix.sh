#!/bin/bash
# Vector instruction mix and lane use of one frame of a bench configuration, on the GPU box (one counter group per process).
# Usage: tools/pmc_mix.sh <tag> "<config> [bench flags]" [lib.so]     -> gpurun_out/pmcmix_<tag>.json
TAG=$1; CFG="$2"
[ -n "$3" ] && export RT_AMD_LIB=$3
OUT=gpurun_out/pmcmix_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="python3 bench.py --config $CFG --no-cpu-baseline --no-strict --steps 1 --warmup 0"
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU --output-format csv -d $OUT/a -- $B > $OUT/a.log 2>&1 || echo "pass a failed"
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR --output-format csv -d $OUT/b -- $B > $OUT/b.log 2>&1 || echo "pass b failed"
python3 - $OUT <<'PY'
import csv, glob, json, sys
out = {}
for f in glob.glob(sys.argv[1] + '/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'render_kernel' in r['Kernel_Name']:
            out[r['Counter_Name']] = out.get(r['Counter_Name'], 0) + float(r['Counter_Value'])
json.dump(out, open(sys.argv[1] + '.json', 'w'), indent=1, sort_keys=True)
print(json.dumps(out, sort_keys=True))
PY
tail -2 $OUT/a.log $OUT/b.log

#!/bin/bash
# One SQ counter pass (wave occupancy, VALU issue, waits) of one frame of a bench configuration, on the GPU box.
# Usage: tools/pmc_sq.sh <tag> "<config> [bench flags]" [lib.so]     -> gpurun_out/pmcsq_<tag>.json
TAG=$1; CFG="$2"
[ -n "$3" ] && export RT_AMD_LIB=$3
OUT=gpurun_out/pmcsq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq -- python3 bench.py --config $CFG --no-cpu-baseline --no-strict --steps 1 --warmup 0 > $OUT/sq.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq2 -- python3 bench.py --config $CFG --no-cpu-baseline --no-strict --steps 1 --warmup 0 > $OUT/sq2.log 2>&1 || exit 1
python3 - $OUT <<'PY'
import csv, glob, json, os, sys
out = {}
for f in glob.glob(sys.argv[1] + '/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if os.environ.get('KFILTER', 'render_kernel') in r['Kernel_Name']:
            out[r['Counter_Name']] = out.get(r['Counter_Name'], 0) + float(r['Counter_Value'])
cyc = out.get('GRBM_GUI_ACTIVE', 0) / 8  # per XCD
out['_valu_util'] = out['SQ_INSTS_VALU'] * 4 / (1024 * cyc) if cyc else None
out['_wait_frac'] = out['SQ_WAIT_ANY'] / out['SQ_WAVE_CYCLES']
out['_cycles'] = cyc
json.dump(out, open(sys.argv[1] + '.json', 'w'), indent=1, sort_keys=True)
print(json.dumps(out, sort_keys=True))
PY

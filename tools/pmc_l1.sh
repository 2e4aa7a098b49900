#!/bin/bash
# Vector L1 (TCP) and address-unit (TA) counters of one frame of the headline bench (GPU box).  Usage: tools/pmc_l1.sh <tag>
set -e
TAG=${1:-l1}
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="python3 bench.py --no-cpu-baseline --no-strict --steps 1 --warmup 0"
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/tcp -- $B > $OUT/tcp.log 2>&1 || echo tcp failed
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum --output-format csv -d $OUT/tcp2 -- $B > $OUT/tcp2.log 2>&1 || echo tcp2 failed
timeout -s KILL 120 rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum --output-format csv -d $OUT/ta -- $B > $OUT/ta.log 2>&1 || echo ta failed
timeout -s KILL 120 rocprofv3 --pmc TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/tcc -- $B > $OUT/tcc.log 2>&1 || echo tcc failed
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

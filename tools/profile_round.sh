#!/bin/bash
# Profiles of the headline bench (run on the GPU box): kernel trace + stats, then PMC passes, each
# in a process of its own (one counter group per pass).  Usage: tools/profile_round.sh r01
set -e
R=${1:-r01}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
export TMPDIR=/tmp
B="python3 bench.py --no-cpu-baseline"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B --steps 3 --warmup 1 > $OUT/trace.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B --steps 1 --warmup 0 --no-strict > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- $B --steps 1 --warmup 0 --no-strict > $OUT/pmc_write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- $B --steps 1 --warmup 0 --no-strict > $OUT/pmc_sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_FLAT --output-format csv -d $OUT/pmc_sq2 -- $B --steps 1 --warmup 0 --no-strict > $OUT/pmc_sq2.log 2>&1 || echo "pmc_sq2 pass failed (non-fatal)"
find $OUT -name "*.csv" | head -50

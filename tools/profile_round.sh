#!/bin/bash
# Profiles of one bench configuration (run on the GPU box): kernel trace + stats, then PMC passes, each in a process
# of its own (one counter group per pass).  Usage: tools/profile_round.sh <tag> [config]      (config: headline|c1..c5)
# Leaves gpurun_out/prof_<tag>/summary/{kernel_stats.csv, pmc_<config>.json, bench_line.json}; copy what is to be kept
# into profiles/<tag>/ and (for the numbers bench.py reads) into profiles/current/.  pmc_<config>.json carries the build id of
# the library it was collected with (rt_build_id); bench.py ignores counters of any other build.
set -e
R=${1:-r02}
CFG=${2:-headline}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
export TMPDIR=/tmp
B="python3 bench.py --config $CFG --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B --steps 3 --warmup 1 > $OUT/trace.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B --steps 1 --warmup 0 --no-strict > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- $B --steps 1 --warmup 0 --no-strict > $OUT/pmc_write.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- $B --steps 1 --warmup 0 --no-strict > $OUT/pmc_sq.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_FLAT --output-format csv -d $OUT/pmc_sq2 -- $B --steps 1 --warmup 0 --no-strict > $OUT/pmc_sq2.log 2>&1 || echo "pmc_sq2 pass failed (non-fatal)"
# vector instruction mix and lane use (two more groups of eight SQ counters)
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU --output-format csv -d $OUT/pmc_mix_a -- $B --steps 1 --warmup 0 --no-strict > $OUT/pmc_mix_a.log 2>&1 || echo "pmc_mix_a pass failed (non-fatal)"
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc_mix_b -- $B --steps 1 --warmup 0 --no-strict > $OUT/pmc_mix_b.log 2>&1 || echo "pmc_mix_b pass failed (non-fatal)"
python3 - $OUT $CFG <<'PY'
# summaries worth committing: per-kernel stats and the counters of the render kernel (summed over the launches of the one frame)
import csv, glob, json, os, sys
out, cfg = sys.argv[1], sys.argv[2]
summ = {}
for grp in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_mix_a", "pmc_mix_b"):
    acc = {}
    for f in glob.glob(f"{out}/{grp}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "render_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    summ[grp] = acc
# which build of the library these counters belong to: bench.py uses them only with that build
import ctypes
lib = ctypes.CDLL(os.environ.get("RT_AMD_LIB") or "cuda_raytracing_amd/librt_amd.so")
lib.rt_build_id.restype = ctypes.c_char_p
summ["build_id"] = lib.rt_build_id().decode()
dst = f"{out}/summary"
os.makedirs(dst, exist_ok=True)
json.dump(summ, open(f"{dst}/pmc_{cfg}.json", "w"), indent=1, sort_keys=True)
for f in glob.glob(f"{out}/trace/*/*_kernel_stats.csv"):
    open(f"{dst}/kernel_stats.csv", "w").write(open(f).read())
for line in open(f"{out}/trace.log"):
    if line.startswith("{"):
        open(f"{dst}/bench_line.json", "w").write(line)
print(open(f"{dst}/kernel_stats.csv").read() if os.path.exists(f"{dst}/kernel_stats.csv") else "no kernel stats")
print(json.dumps(summ, indent=1, sort_keys=True))
PY

#!/bin/bash
# Profiles bench.py on the GPU box: kernel trace + stats, then PMC passes (each in its own run).
# Usage: tools/profile_bench.sh <tag>     -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r1}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# PROFILE_CMD replaces the benchmark command, e.g. PROFILE_CMD="python3 $PWD/tools/run_configs.py c5m"
BENCH=${PROFILE_CMD:-"python3 $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"}
cd /tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU --output-format csv -d "$OUT/pmc_sq" -- $BENCH > "$OUT/pmc_sq.log" 2>&1 || { echo "pmc_sq failed"; tail -5 "$OUT/pmc_sq.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_VMEM GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq2" -- $BENCH > "$OUT/pmc_sq2.log" 2>&1 || { echo "pmc_sq2 failed"; tail -5 "$OUT/pmc_sq2.log"; }
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/pmc_fetch.log" 2>&1 || { echo "pmc_fetch failed"; tail -5 "$OUT/pmc_fetch.log"; }
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/pmc_write.log" 2>&1 || { echo "pmc_write failed"; tail -5 "$OUT/pmc_write.log"; }
find "$OUT" -name "*.csv" | head -30

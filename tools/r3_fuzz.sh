#!/bin/bash
# Long randomised parity runs against the oracle (GPU box): large scenes (hierarchies in LDS and in global memory), then small ones.
# Usage: bash tools/r3_fuzz.sh BIG_FIRST BIG_COUNT SMALL_FIRST SMALL_COUNT   -> gpurun_out/r3c/fuzz_*.log
set -e
O=gpurun_out/r3c
mkdir -p $O
timeout -k 10 1000 python -u tools/fuzz_many.py $1 $2 big > $O/fuzz_big_$2.log 2>&1 || { tail -5 $O/fuzz_big_$2.log; exit 1; }
tail -1 $O/fuzz_big_$2.log
timeout -k 10 600 python -u tools/fuzz_many.py $3 $4 > $O/fuzz_$4.log 2>&1 || { tail -5 $O/fuzz_$4.log; exit 1; }
tail -1 $O/fuzz_$4.log

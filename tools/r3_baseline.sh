#!/bin/bash
# Baseline on the GPU box: the whole GPU suite, the bench line, and the rocprofv3 passes of the benchmark.
# Results: gpurun_out/r3b/ and gpurun_out/prof_r3b/.
set -e
O=gpurun_out/r3b
mkdir -p $O
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
timeout -k 10 300 python bench.py > $O/bench_line.json 2> $O/bench_err.log || { tail -20 $O/bench_err.log; exit 1; }
cat $O/bench_line.json
bash tools/profile_bench.sh r3b
python tools/summarize_profile.py r3b > $O/summary.log 2>&1 || tail -5 $O/summary.log
cp -r profiles/r3b $O/profiles_r3b; tail -20 $O/summary.log

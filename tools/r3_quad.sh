#!/bin/bash
# The four-way tree walk against the depth-first walk: parity tests, then timings (GPU box).
set -e
O=gpurun_out/r3c
mkdir -p $O
timeout -k 10 900 python -u -m pytest tests/test_hierarchy.py tests/test_fuzz_gpu.py tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest_quad.log 2>&1 || { tail -30 $O/pytest_quad.log; exit 1; }
tail -3 $O/pytest_quad.log
p() { python -c "import json,sys; d=json.load(open('$1')); print('$2', 'ms_per_step %.3f value %.4e kernel_ms %.3f' % (d['ms_per_step'], d['value'], d['roofline']['kernel_ms']))"; }
for round in 1 2; do
  timeout -k 10 200 python bench.py --workload c5 --frames 8 --steps 6 --warmup 2 --no-cpu-baseline > $O/c5_quad.json 2> $O/err.log; p $O/c5_quad.json "c5 8 frames, four-way tree"
  RT_TREE_DEPTH_FIRST=1 timeout -k 10 200 python bench.py --workload c5 --frames 8 --steps 6 --warmup 2 --no-cpu-baseline > $O/c5_dfs.json 2> $O/err.log; p $O/c5_dfs.json "c5 8 frames, depth-first tree"
done
timeout -k 10 300 python tools/run_configs.py c5 c5m rnd | tee $O/configs_quad.txt
RT_TREE_DEPTH_FIRST=1 timeout -k 10 300 python tools/run_configs.py c5 c5m rnd | tee $O/configs_dfs.txt

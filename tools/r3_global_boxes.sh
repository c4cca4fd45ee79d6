#!/bin/bash
# The box tree walked in global memory (scenes beyond about 6000 spheres): parity tests, then timings against the depth-first tree of balls (GPU box).
set -e
O=gpurun_out/r3e
mkdir -p $O
timeout -k 10 900 python -u -m pytest tests/test_hierarchy.py tests/test_fuzz_gpu.py tests/test_gpu_configs.py -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
timeout -k 10 600 python tools/large_scene_timing.py 4096 5000 8192 16384 65536 262144 1000000 | tee $O/large_scene_timing.txt
cp gpurun_out/large_scene_timing.json $O/
timeout -k 10 900 python -u tools/fuzz_many.py 920000 4000 big > $O/fuzz_big_4000.log 2>&1 || { tail -5 $O/fuzz_big_4000.log; exit 1; }
tail -1 $O/fuzz_big_4000.log

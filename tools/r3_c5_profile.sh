#!/bin/bash
# C5 (4096 spheres, depth 50, 8 frames) on the GPU box: bench line, section counters of the diagnostic build, rocprofv3 passes.
set -e
O=gpurun_out/r3h
mkdir -p $O
timeout -k 10 200 python bench.py --workload c5 --frames 8 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c5.json 2> $O/err.log || { tail $O/err.log; exit 1; }
python -c "import json; d=json.load(open('$O/bench_c5.json')); print('c5 ms_per_step %.3f kernel_ms %.3f frac %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
bash tools/r3_stats.sh
PROFILE_CMD="python3 $PWD/bench.py --workload c5 --frames 8 --steps 3 --warmup 1 --no-cpu-baseline" bash tools/profile_bench.sh r3h_c5
python tools/summarize_profile.py r3h_c5 $((1920*1080*16*8)) > $O/summary_c5.log 2>&1 || tail -5 $O/summary_c5.log
mkdir -p $O/profiles_r3h_c5 && cp profiles/r3h_c5/* $O/profiles_r3h_c5/
tail -16 $O/summary_c5.log
timeout -k 10 300 python tools/run_configs.py c5 c5m rnd | tee $O/configs_boxes.txt

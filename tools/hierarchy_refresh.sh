#!/bin/bash
# Figures of the hierarchy kernels after a change of the box walk (GPU box): C5 at 1 / 8 / 32 / 64 frames per launch, rank 0's
# part of an eight-rank frame split, C5m and the random scene, the synthetic grid from 4096 to a million spheres, the section
# counters of the diagnostic build and the rocprofv3 passes of the 8-frame launch.   Usage: bash tools/hierarchy_refresh.sh <tag>
set -e
TAG=${1:-hierarchy}
O=gpurun_out/$TAG
mkdir -p $O
p() { python -c "import json,sys; d=json.load(open('$1')); print('$2', 'ms_per_step %.3f value %.4e kernel_ms %.3f' % (d['ms_per_step'], d['value'], d['roofline']['kernel_ms']))"; }
for f in 8 32 64; do
  timeout -k 10 200 python bench.py --workload c5 --frames $f --steps 6 --warmup 2 --no-cpu-baseline > $O/c5_f${f}_w1.json 2> $O/err.log; p $O/c5_f${f}_w1.json "c5 one GPU, $f frames"
  timeout -k 10 200 python bench.py --workload c5 --mode frames --frames $f --rehearse-world 8 --steps 12 --warmup 3 --no-cpu-baseline > $O/c5_f${f}_w8.json 2> $O/err.log; p $O/c5_f${f}_w8.json "c5 rank 0 of 8, $f frames"
done
timeout -k 10 300 python tools/run_configs.py c5 c5m rnd | tee $O/configs_boxes.txt
timeout -k 10 600 python tools/large_scene_timing.py 4096 5000 8192 16384 65536 262144 1000000 | tee $O/large_scene_timing.txt
cp gpurun_out/large_scene_timing.json $O/
RT_STATS_SCENE=synthetic4096 RT_STATS_SIZE="1920 1080" timeout -k 10 300 python tools/section_stats.py 8 > $O/section_stats_c5_boxes.txt 2> $O/err.log; cat $O/section_stats_c5_boxes.txt
PROFILE_CMD="python3 $PWD/bench.py --workload c5 --frames 8 --steps 3 --warmup 1 --no-cpu-baseline" bash tools/profile_bench.sh ${TAG}_c5
python tools/summarize_profile.py ${TAG}_c5 $((1920*1080*16*8)) > $O/summary_c5.log 2>&1 || tail -5 $O/summary_c5.log
mkdir -p $O/profiles_${TAG}_c5 && cp profiles/${TAG}_c5/* $O/profiles_${TAG}_c5/
tail -16 $O/summary_c5.log

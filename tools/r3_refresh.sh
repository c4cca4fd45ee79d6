#!/bin/bash
# Refreshes the C5 and large-scene figures after a change of the hierarchy walk (GPU box): profile, frames-per-step rehearsal, scene sizes.
set -e
bash tools/r3_c5_profile.sh
bash tools/r3_c5_frames.sh
timeout -k 10 600 python tools/large_scene_timing.py 4096 5000 8192 16384 65536 262144 1000000 | tee gpurun_out/r3h/large_scene_timing.txt
cp gpurun_out/large_scene_timing.json gpurun_out/r3h/

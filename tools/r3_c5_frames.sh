#!/bin/bash
# C5 (4096 spheres, depth 50) with more frames per step: one GPU, and rank 0's part of an eight-rank frame split
# (bench.py --rehearse-world 8), for 8 / 32 / 64 frames per step; and the unit size of multi-frame hierarchy
# launches (units of 8 pixels x 4 frames by default) against larger units.
# Results: gpurun_out/r3a/*.json.   Usage (GPU box): bash tools/r3_c5_frames.sh
set -e
O=gpurun_out/r3h
mkdir -p $O
p() { python -c "import json,sys; d=json.load(open('$1')); print('$2', 'ms_per_step %.3f value %.4e kernel_ms %.3f' % (d['ms_per_step'], d['value'], d['roofline']['kernel_ms']))"; }
for f in 8 32 64; do
  timeout -k 10 200 python bench.py --workload c5 --frames $f --steps 6 --warmup 2 --no-cpu-baseline > $O/c5_f${f}_w1.json 2> $O/err.log; p $O/c5_f${f}_w1.json "c5 one GPU, $f frames"
  timeout -k 10 200 python bench.py --workload c5 --mode frames --frames $f --rehearse-world 8 --steps 12 --warmup 3 --no-cpu-baseline > $O/c5_f${f}_w8.json 2> $O/err.log; p $O/c5_f${f}_w8.json "c5 rank 0 of 8, $f frames"
done
for items in 128 512; do
  for f in 16 64; do
    RT_UNIT_ITEMS=$items timeout -k 10 200 python bench.py --workload c5 --frames $f --steps 4 --warmup 1 --no-cpu-baseline > $O/c5_f${f}_items${items}.json 2> $O/err.log; p $O/c5_f${f}_items${items}.json "c5 one GPU, $f frames, RT_UNIT_ITEMS=$items"
  done
done
timeout -k 10 200 python bench.py --workload c5 --frames 16 --steps 4 --warmup 1 --no-cpu-baseline > $O/c5_f16_w1.json 2> $O/err.log; p $O/c5_f16_w1.json "c5 one GPU, 16 frames"

#!/bin/bash
# Section counters of the hierarchy kernel on the 4096-sphere scene, both walks (GPU box, diagnostic build).
set -e
O=gpurun_out/r3h
mkdir -p $O
RT_STATS_SCENE=synthetic4096 RT_STATS_SIZE="1920 1080" timeout -k 10 300 python tools/section_stats.py 8 > $O/section_stats_c5_quad.txt 2> $O/err.log; cat $O/section_stats_c5_quad.txt
cp gpurun_out/section_stats.json $O/section_stats_c5_quad.json
RT_TREE_DEPTH_FIRST=1 RT_STATS_SCENE=synthetic4096 RT_STATS_SIZE="1920 1080" timeout -k 10 300 python tools/section_stats.py 8 > $O/section_stats_c5_dfs.txt 2> $O/err.log; cat $O/section_stats_c5_dfs.txt

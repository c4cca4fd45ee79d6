#!/bin/bash
# Timing rehearsals of the multi-GPU step loop on a one-GPU box: this process does rank 0's part of a job of 2, 4
# and 8 ranks (bench.py --rehearse-world: the collective is replaced by device copies of the same size), for the
# benchmark workload in both split modes, the 8K frame and the 4096-sphere scene; then the one-rank RCCL runs
# and the multi-process GPU tests.  Results: gpurun_out/<tag>/*.json.   Usage (GPU box): bash tools/rehearse_all.sh <tag>
set -e
O=gpurun_out/${1:-rehearse}
mkdir -p $O
p() { python -c "import json,sys; d=json.load(open('$1')); print('$2', 'ms_per_step %.3f value %.3e' % (d['ms_per_step'], d['value']))"; }
for w in 8 4 2; do
  timeout -k 10 120 python bench.py --rehearse-world $w --steps 100 --warmup 10 --no-cpu-baseline > $O/rehearse_w${w}_p0.json 2> $O/err.log; p $O/rehearse_w${w}_p0.json "c2 tiles world $w"
done
timeout -k 10 200 python bench.py --workload c4 --rehearse-world 8 --steps 20 --warmup 3 --no-cpu-baseline > $O/rehearse_c4_w8.json 2> $O/err.log; p $O/rehearse_c4_w8.json "c4 tiles world 8"
timeout -k 10 200 python bench.py --workload c5 --frames 8 --mode frames --rehearse-world 8 --steps 20 --warmup 3 --no-cpu-baseline > $O/rehearse_c5_w8.json 2> $O/err.log; p $O/rehearse_c5_w8.json "c5 frames world 8"
timeout -k 10 200 python bench.py --mode frames --rehearse-world 8 --steps 50 --warmup 5 --no-cpu-baseline > $O/rehearse_c2_frames_w8.json 2> $O/err.log; p $O/rehearse_c2_frames_w8.json "c2 frames world 8"
for m in tiles frames; do
  timeout -k 10 200 python bench.py --force-distributed --mode $m --steps 20 --warmup 5 --no-cpu-baseline > $O/force_distributed_$m.json 2> $O/err.log; p $O/force_distributed_$m.json "c2 $m one-rank RCCL"
done
timeout -k 10 400 python -u -m pytest tests/test_gpu_dist.py -m gpu -x -q --timeout 200 --timeout-method thread 2>&1 | tail -3

#!/bin/bash
# The native RCCL path of bench.py on a one-rank communicator, both split modes, verified against one GPU's image (GPU box).
set -e
O=gpurun_out/r3b
mkdir -p $O
p() { python -c "import json,sys; d=json.load(open('$1')); print('$2', 'ms_per_step %.3f value %.4e' % (d['ms_per_step'], d['value']), d.get('verified_against_one_gpu'), d['config']['parallelism'][-60:])"; }
for m in tiles frames; do
  for c in native torch; do
    timeout -k 10 200 python bench.py --force-distributed --mode $m --collectives $c --steps 20 --warmup 5 --no-cpu-baseline --verify > $O/force_distributed_${m}_${c}.json 2> $O/err.log || { tail -20 $O/err.log; exit 1; }
    p $O/force_distributed_${m}_${c}.json "c2 $m one-rank $c"
  done
done
timeout -k 10 200 python bench.py --workload c5 --frames 8 --mode frames --force-distributed --steps 6 --warmup 2 --no-cpu-baseline --verify > $O/force_distributed_c5_native.json 2> $O/err.log || { tail -20 $O/err.log; exit 1; }
p $O/force_distributed_c5_native.json "c5 frames one-rank native"
timeout -k 10 200 python bench.py --workload c4 --mode tiles --force-distributed --steps 6 --warmup 2 --no-cpu-baseline --verify > $O/force_distributed_c4_native.json 2> $O/err.log || { tail -20 $O/err.log; exit 1; }
p $O/force_distributed_c4_native.json "c4 tiles one-rank native"

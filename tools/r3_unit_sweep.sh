#!/bin/bash
# Unit size of hierarchy launches on the box tree (GPU box): C5 with 1, 8 and 32 frames per launch, RT_UNIT_ITEMS swept.
set -e
O=gpurun_out/r3f
mkdir -p $O
p() { python -c "import json,sys; d=json.load(open('$1')); print('$2', 'ms_per_step %.3f' % d['ms_per_step'])"; }
for f in 1 8 32; do
  for items in 0 16 32 64 128 256; do
    RT_UNIT_ITEMS=$items timeout -k 10 200 python bench.py --workload c5 --frames $f --steps 6 --warmup 2 --no-cpu-baseline > $O/c5_f${f}_items${items}.json 2> $O/err.log; p $O/c5_f${f}_items${items}.json "c5 $f frames, RT_UNIT_ITEMS=$items"
  done
done

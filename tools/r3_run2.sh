#!/bin/bash
# Measurements on the GPU box: contract-cost report, unit-size sweep of multi-frame hierarchy launches.
set -e
O=gpurun_out/r3a
mkdir -p $O
timeout -k 10 300 python tools/contract_cost.py 1280x720 > $O/contract_cost.json 2> $O/err.log; cat $O/contract_cost.json
bash tools/ab_bench.sh ray-tracing_amd/librt_mi355x.so ray-tracing_amd/variants/librt_mi355x_contract.so | tee $O/contract_ab.txt
RT_MI355X_LIB=$PWD/ray-tracing_amd/variants/librt_mi355x_contract.so timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_contract.json 2>> $O/err.log
p() { python -c "import json,sys; d=json.load(open('$1')); print('$2', 'ms_per_step %.3f value %.4e kernel_ms %.3f' % (d['ms_per_step'], d['value'], d['roofline']['kernel_ms']))"; }
p $O/bench_contract.json "c2 contract build"
for items in 32 64 128 256; do
  for f in 8 16 32; do
    RT_UNIT_ITEMS=$items timeout -k 10 200 python bench.py --workload c5 --frames $f --steps 4 --warmup 1 --no-cpu-baseline > $O/c5_f${f}_items${items}.json 2>> $O/err.log; p $O/c5_f${f}_items${items}.json "c5 one GPU, $f frames, RT_UNIT_ITEMS=$items"
  done
done

#!/bin/bash
# A/B timing of library variants on the GPU box: tools/ab_bench.sh <lib.so> [<lib.so> ...]
# Runs bench.py (5 steps) per variant, two rounds interleaved, prints Gsamples/s and kernel ms.
# AB_ARGS adds bench.py arguments, e.g. AB_ARGS="--workload c5".
for round in 1 2; do
  for lib in "$@"; do
    RT_MI355X_LIB=$PWD/$lib timeout -k 10 ${AB_TIMEOUT:-200} python bench.py $AB_ARGS --steps 5 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-40s %.2f Gs/s  kernel %.2f ms  step %.2f ms' % ('$lib', d['value']/1e9, d['roofline']['kernel_ms'], d['ms_per_step']))" || exit 1
  done
done

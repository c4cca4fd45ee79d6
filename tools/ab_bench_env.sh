#!/bin/bash
# A/B timing with an environment setting per run: tools/ab_bench_env.sh "VAR=val lib.so" ["VAR=val lib.so" ...]
for round in 1 2; do
  for spec in "$@"; do
    setting=${spec%% *}
    lib=${spec##* }
    env $setting RT_MI355X_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-50s %.2f Gs/s  kernel %.2f ms' % ('$spec', d['value']/1e9, d['roofline']['kernel_ms']))" || exit 1
  done
done

#!/bin/bash
# A/B timing with environment settings per run.  Each argument is "VAR=val[,VAR2=val2...] lib.so";
# an underscore in a value stands for a space.   tools/ab_bench_env.sh "RT_TILE_LOG2=3_3,RT_FRAMES_PER_UNIT=8 lib.so" ...
for round in 1 2; do
  for spec in "$@"; do
    settings=${spec%% *}
    lib=${spec##* }
    envargs=()
    IFS=',' read -ra parts <<< "$settings"
    for kv in "${parts[@]}"; do name=${kv%%=*}; value=${kv#*=}; envargs+=("$name=${value//_/ }"); done
    env "${envargs[@]}" RT_MI355X_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-60s %.2f Gs/s  kernel %.2f ms' % ('$spec', d['value']/1e9, d['roofline']['kernel_ms']))" || exit 1
  done
done

#!/bin/bash
# A/B timing of library variants on scenes that go through the hierarchy (GPU box): tools/ab_scenes.sh <lib.so> [<lib.so> ...]
# Per variant: the 485-sphere random scene, C5 (one frame) and C5m (tools/run_configs.py), then the synthetic grid at
# AB_COUNTS spheres (tools/large_scene_timing.py), two interleaved rounds.
for round in 1 2; do
  for lib in "$@"; do
    echo "== $lib (round $round)"
    RT_MI355X_LIB=$PWD/$lib timeout -k 10 300 python tools/run_configs.py rnd c5 c5m 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    name, rest = line.split(' ', 1)
    print('  %-6s kernel %.3f ms' % (name, json.loads(rest)['kernel_ms']))" || exit 1
    RT_MI355X_LIB=$PWD/$lib timeout -k 10 300 python tools/large_scene_timing.py ${AB_COUNTS:-1024 16384 262144} 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    name, rest = line.split(' ', 1)
    d = json.loads(rest)
    print('  %-8s hierarchy %.3f ms identical %s' % (name, d['hierarchy_ms'], d['identical']))" || exit 1
  done
done

#!/bin/bash
# Whole-tree check on the GPU box: smoke, the whole GPU suite, the bench line, the one-rank native collective path, C5 line,
# and the rocprofv3 passes of the benchmark command.   Usage: bash tools/gpu_check.sh <tag>
set -e
O=gpurun_out/${1:-check}
mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
timeout -k 10 300 python bench.py > $O/bench_line.json 2> $O/bench_err.log || { tail -20 $O/bench_err.log; exit 1; }
python -c "import json; d=json.load(open('$O/bench_line.json')); print('c2', d['value'], d['ms_per_step'], d['roofline']['frac'])"
timeout -k 10 200 python bench.py --workload c5 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_c5.json 2> $O/err.log || { tail $O/err.log; exit 1; }
python -c "import json; d=json.load(open('$O/bench_c5.json')); print('c5', d['value'], d['ms_per_step'], d['roofline']['frac'])"
for m in tiles frames; do
  timeout -k 10 200 python bench.py --force-distributed --mode $m --steps 20 --warmup 5 --no-cpu-baseline --verify > $O/force_distributed_${m}_native.json 2> $O/err.log || { tail -20 $O/err.log; exit 1; }
  python -c "import json; d=json.load(open('$O/force_distributed_${m}_native.json')); print('$m', d['value'], d.get('verified_against_one_gpu'), d['config']['parallelism'][-90:])"
done
timeout -k 10 200 python bench.py --workload c5 --mode frames --force-distributed --steps 4 --warmup 1 --no-cpu-baseline --verify > $O/force_distributed_c5_native.json 2> $O/err.log || { tail -20 $O/err.log; exit 1; }
python -c "import json; d=json.load(open('$O/force_distributed_c5_native.json')); print('c5 frames', d['value'], d.get('verified_against_one_gpu'))"
bash tools/profile_bench.sh ${1:-check}
python tools/summarize_profile.py ${1:-check} > $O/summary.log 2>&1 || tail -5 $O/summary.log
mkdir -p $O/profiles && cp profiles/${1:-check}/* $O/profiles/
tail -14 $O/summary.log

# single-frame launches of the 4096-sphere scene (what a rank of the eight-rank frame split runs): unit geometry sweep
set -e
O=gpurun_out/${1:-rehearse}
mkdir -p $O
for tile in "default" "2 3" "2 2" "1 2" "3 3" "1 1"; do
  if [ "$tile" = default ]; then unset RT_TILE_LOG2; else export RT_TILE_LOG2="$tile"; fi
  timeout -k 10 200 python bench.py --workload c5 --frames 8 --mode frames --rehearse-world 8 --steps 20 --warmup 3 --no-cpu-baseline > $O/c5_tile.json 2> $O/c5_tile.err
  python - <<PY
import json
d=json.load(open("$O/c5_tile.json"))
print("tile log2 [$tile]: ms_per_step %.3f kernel_ms %.3f" % (d["ms_per_step"], d["roofline"]["kernel_ms"]))
PY
done

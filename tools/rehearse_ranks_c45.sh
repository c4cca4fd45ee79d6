set -e
mkdir -p gpurun_out/r2g
for spec in "c4 tiles" "c5 frames"; do
  set -- $spec
  for w in 1 8; do
    extra="--rehearse-world $w"; [ $w = 1 ] && extra="--force-distributed"
    timeout -k 10 200 python bench.py --workload $1 --mode $2 $extra --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2g/rehearse_$1_w$w.json 2> gpurun_out/r2g/rehearse_$1_w$w.err
    python - <<PY
import json
d=json.load(open("gpurun_out/r2g/rehearse_$1_w$w.json"))
print("$1 $2 world $w: ms_per_step %.3f value %.3e kernel_ms %.3f" % (d["ms_per_step"], d["value"], d["roofline"]["kernel_ms"]))
PY
  done
done

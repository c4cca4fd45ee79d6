set -e
mkdir -p gpurun_out/r2g
for pr in 0 -1; do
  for w in 8 4 2; do
    timeout -k 10 120 python bench.py --rehearse-world $w --comm-priority $pr --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r2g/rehearse_w${w}_p${pr}.json 2> gpurun_out/r2g/rehearse_w${w}_p${pr}.err
    python - <<PY
import json
d=json.load(open("gpurun_out/r2g/rehearse_w${w}_p${pr}.json"))
print("world $w prio $pr: ms_per_step %.3f value %.3e kernel_ms %.3f" % (d["ms_per_step"], d["value"], d["roofline"]["kernel_ms"]))
PY
  done
done

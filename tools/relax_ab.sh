set -e
bash tools/ab_bench.sh ray-tracing_amd/librt_mi355x.so ray-tracing_amd/librt_relax3.so ray-tracing_amd/librt_relax4.so
for lib in librt_mi355x.so librt_relax3.so librt_relax4.so; do
  echo "== $lib"
  RT_MI355X_LIB=$PWD/ray-tracing_amd/$lib timeout -k 10 200 python tools/overlap_probe.py 16 8 0 2>/dev/null | head -2
  RT_MI355X_LIB=$PWD/ray-tracing_amd/$lib timeout -k 10 200 python tools/overlap_probe.py 4 12 0.05 2>/dev/null | head -2
done

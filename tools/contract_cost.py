#!/usr/bin/env python3
"""What the bit-exact contract costs, measured: the shipped library (every operation rounded once, no
multiply-add contraction) against a build of the SAME sources with -ffp-contract=fast
(`make -C ray-tracing_amd/csrc contract`).  Renders 64 frames of 64 spp (4096 spp, depth 16, aperture 0.05)
with both, frame by frame with the final RNG words captured, and reports the fraction of (pixel, frame) items
whose random stream diverged and the RMSE of the converged radiance relative to the mean luminance.
Throughput of the two builds: bench.py with RT_MI355X_LIB (tools/ab_bench.sh).  Report only: the contracted
build is never shipped or loaded by default.   Usage (GPU box): python tools/contract_cost.py [WxH]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch  # noqa: F401  (its HIP runtime first)
    import rtamd_loader
    pkg = rtamd_loader.load()
    from raytracing_amd import image, tracer
    w, h = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1280x720").split("x"))
    spp, depth, frames = 64, 16, 64
    fast = tracer.load_library(os.path.join(ROOT, "ray-tracing_amd", "variants", "librt_mi355x_contract.so"))
    t = {"strict": pkg.RayTracer(), "contract": pkg.RayTracer(lib=fast)}
    for x in t.values():
        x.create(w, h, device_id=0)
        x.set_samples_per_pixel(spp)
        x.set_max_depth(depth)
        x.set_aperture(0.05)
        x.enable_rng_capture(True)
    diverged = 0
    for f in range(frames):
        for x in t.values():
            x.render(f)
        diverged += int(np.count_nonzero(t["strict"].read_rng_state() != t["contract"].read_rng_state()))
    a, b = t["strict"].read_accum(), t["contract"].read_accum()
    rmse, rel = image.rmse_linear(b, frames * spp, a, frames * spp)
    out = {"image": "%dx%d, default scene and camera, %d spp x %d frames, depth %d, aperture 0.05" % (w, h, spp, frames, depth),
           "items": w * h * frames, "diverged_items": diverged, "diverged_fraction": diverged / float(w * h * frames),
           "rmse": rmse, "rmse_over_mean_luminance": rel,
           "bit_identical_pixels": int(np.count_nonzero(np.all(a.view(np.uint32) == b.view(np.uint32), axis=2))),
           "pixels": w * h}
    print(json.dumps(out, indent=1))
    for x in t.values():
        x.destroy()


if __name__ == "__main__":
    main()

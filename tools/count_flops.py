#!/usr/bin/env python3
"""Counts the algorithmic FLOP per path sample of a workload with the instrumented CPU oracle and
records it in profiles/flops_per_sample.json (bench.py's roofline line uses it).

Counting convention: oracle/pyoracle.py:flops_from_counters.  Usage:
    python tools/count_flops.py [--frames N]      (default: frames 0..7 of the 1080p workload)
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    args = ap.parse_args()
    import rtamd_loader
    from oracle import pyoracle
    pkg = rtamd_loader.load()
    host = pkg.RayTracer().create(1920, 1080, device_id=-1)
    host.set_samples_per_pixel(16)
    host.set_max_depth(8)
    host.set_aperture(0.0)
    scene = pkg.default_scene()
    total = {k: 0 for k in pyoracle.COUNTER_FIELDS}
    for f in range(args.frames):
        r = pyoracle.render(host.get_params_words(f), scene, want_output=False, want_counters=True)
        for k, v in r["counters"].items():
            total[k] += v
        print("frame %d done" % f, flush=True)
    flops = pyoracle.flops_from_counters(total)
    rec = {
        "flops_per_sample": flops / total["samples"],
        "segments_per_sample": total["segments"] / total["samples"],
        "rand_draws_per_sample": total["rand_draws"] / total["samples"],
        "frames_counted": args.frames,
        "counters": total,
        "convention": "add/sub/mul/div/sqrt/inversesqrt/floor/pow = 1; compare/select/neg/min/max/int/convert = 0; "
                      "no FMA credit; counted as written in shaders/raytrace.comp.glsl",
    }
    path = os.path.join(ROOT, "profiles", "flops_per_sample.json")
    data = {}
    if os.path.exists(path):
        data = json.load(open(path))
    data["c2_1080p_spp16_depth8_dof_off"] = rec
    json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps({k: rec[k] for k in ("flops_per_sample", "segments_per_sample", "rand_draws_per_sample")}))


if __name__ == "__main__":
    main()

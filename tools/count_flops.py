#!/usr/bin/env python3
"""Counts the algorithmic FLOP per path sample of a workload with the instrumented CPU oracle and
records it in profiles/flops_per_sample.json (bench.py's roofline line uses it).

Counting convention: oracle/pyoracle.py:flops_from_counters (operations as written in the shader; for the
4096-sphere scene that is the loop over all spheres, 23 FLOP per sphere and segment, whatever the tracer
does to avoid most of those tests).  Usage:
    python tools/count_flops.py [--workload c2|c3|c4|c5] [--frames N] [--band-step K]
Large workloads are counted on a sample: 8-row bands, every K-th band of the image.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

DEFAULTS = {"c2": (8, 1), "c3": (1, 1), "c4": (1, 8), "c5": (1, 45)}   # frames counted, band step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--frames", type=int, default=0)
    ap.add_argument("--band-step", type=int, default=0)
    args = ap.parse_args()
    import bench
    import rtamd_loader
    from oracle import pyoracle
    pkg = rtamd_loader.load()
    wl = bench.WORKLOADS[args.workload]
    frames = args.frames or DEFAULTS[args.workload][0]
    step = args.band_step or DEFAULTS[args.workload][1]
    host = pkg.RayTracer().create(wl["width"], wl["height"], device_id=-1)
    host.set_samples_per_pixel(wl["spp"])
    host.set_max_depth(wl["depth"])
    host.set_aperture(wl["aperture"])
    scene = pkg.default_scene()
    if wl["scene"] == "synthetic4096":
        scene = pkg.synthetic_scene(4096)
        host.set_spheres(scene)
        host.set_camera([3.0, 1.5, 6.0], [-0.4, -0.2, -1.0], 6.0)
    bands = [(y, min(y + 8, wl["height"])) for y in range(0, wl["height"], 8)][step // 2::step]
    total = {k: 0 for k in pyoracle.COUNTER_FIELDS}
    accum = None
    for f in range(frames):
        for (y0, y1) in bands:
            r = pyoracle.render(host.get_params_words(f), scene, accum=accum, y0=y0, y1=y1, want_output=False,
                                want_counters=True)
            accum = r["accum"]
            for k, v in r["counters"].items():
                total[k] += v
        print("frame %d done" % f, flush=True)
    flops = pyoracle.flops_from_counters(total)
    rec = {
        "flops_per_sample": flops / total["samples"],
        "segments_per_sample": total["segments"] / total["samples"],
        "rand_draws_per_sample": total["rand_draws"] / total["samples"],
        "frames_counted": frames,
        "rows_counted": "every %d-th band of 8 rows (%d of %d rows)" % (step, sum(b - a for a, b in bands), wl["height"]),
        "counters": total,
        "convention": "add/sub/mul/div/sqrt/inversesqrt/floor/pow = 1; compare/select/neg/min/max/int/convert = 0; "
                      "no FMA credit; counted as written in shaders/raytrace.comp.glsl",
    }
    path = os.path.join(ROOT, "profiles", "flops_per_sample.json")
    data = {}
    if os.path.exists(path):
        data = json.load(open(path))
    data[wl["flops_key"]] = rec
    json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps({k: rec[k] for k in ("flops_per_sample", "segments_per_sample", "rand_draws_per_sample")}))


if __name__ == "__main__":
    main()

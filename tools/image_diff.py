#!/usr/bin/env python3
"""Compares two renders.

  python tools/image_diff.py a.png b.png          display images (PNG/PPM): RMSE / max / differing fraction in code values
  python tools/image_diff.py a.npz b.npz          accumulators saved by tools/render.py --accum: RMSE in linear radiance,
                                                  absolute and relative to the mean luminance
  options: --max-rmse X   exit status 1 if the (relative, for accumulators) RMSE exceeds X
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _read_display(image, path):
    return image.read_ppm(path) if path.lower().endswith(".ppm") else image.read_png(path)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("a")
    ap.add_argument("b")
    ap.add_argument("--max-rmse", type=float)
    args = ap.parse_args()
    import rtamd_loader
    rtamd_loader.load()
    from raytracing_amd import image

    if args.a.endswith(".npz") != args.b.endswith(".npz"):
        ap.error("compare two accumulators or two display images, not one of each")
    if args.a.endswith(".npz"):
        acc_a, frames_a, spp_a = image.load_accum(args.a)
        acc_b, frames_b, spp_b = image.load_accum(args.b)
        rmse, rel = image.rmse_linear(acc_a, frames_a * spp_a, acc_b, frames_b * spp_b)
        report = {"space": "linear", "rmse": rmse, "relative_rmse": rel}
        measure = rel
    else:
        report = {"space": "rgba8"}
        report.update(image.diff_rgba8(_read_display(image, args.a), _read_display(image, args.b)))
        measure = report["rmse"]
    print(json.dumps(report))
    if args.max_rmse is not None and measure > args.max_rmse:
        sys.exit(1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""How much of the cost of single-frame launches is the launch tail: renders single frames of the benchmark
workload through K handles on K streams (frames dealt round-robin) and prints the time per frame.
Usage (GPU box): python tools/overlap_probe.py [spp depth aperture]"""
import sys
import os
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import rtamd_loader
    pkg = rtamd_loader.load()
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    depth = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    aperture = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    frames = 128
    others = [pkg.RayTracer().create(64, 64, device_id=0) for _ in range(int(os.environ.get("PROBE_OTHER_HANDLES", "0")))]
    t = pkg.RayTracer().create(1920, 1080, device_id=0)
    t.set_samples_per_pixel(spp)
    t.set_max_depth(depth)
    t.set_aperture(aperture)
    for overlap in (False, True):
        t.set_frame_overlap(overlap)
        for rep in range(2):
            t.synchronize()
            t0 = time.perf_counter()
            for f in range(frames):
                t.render(f)
            t.synchronize()
            dt = time.perf_counter() - t0
        print("one handle, library overlap %s: %.3f ms per frame, %.2f Gsamples/s" % (
            "on" if overlap else "off", dt / frames * 1e3, 1920 * 1080 * spp * frames / dt / 1e9), flush=True)
    t.destroy()
    for k in (1, 2, 3, 4):
        streams = [torch.cuda.Stream() for _ in range(k)]
        tracers = []
        for s in streams:
            t = pkg.RayTracer().create(1920, 1080, device_id=0)
            t.set_samples_per_pixel(spp)
            t.set_max_depth(depth)
            t.set_aperture(aperture)
            t.set_stream(s.cuda_stream)
            t.set_frame_overlap(False)
            tracers.append(t)
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for f in range(frames):
                tracers[f % k].render(f)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print("%d stream(s): %.3f ms per frame, %.2f Gsamples/s" % (k, dt / frames * 1e3, 1920 * 1080 * spp * frames / dt / 1e9),
              flush=True)
        for t in tracers:
            t.destroy()


if __name__ == "__main__":
    main()

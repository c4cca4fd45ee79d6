#!/usr/bin/env python3
"""What the small-scene kernel reaches when the lanes of a wave do the same thing (GPU box): the benchmark's settings
(1920x1080, 16 spp, depth 8, 64 frames, the default scene) with the camera turned so that every path is alike --
"sky": straight up, every ray leaves the scene at once; "ground": from high above straight down onto the checkered
ground away from the spheres, every path is one lambert bounce and then the sky -- next to the benchmark's own view.
Algorithmic FLOP per sample of each view come from the instrumented CPU oracle (a 240x135 image of the same view);
the kernel time from the library's HIP events.  python tools/uniform_lanes_probe.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VIEWS = {
    "benchmark": None,
    "sky": ([13.0, 2.0, 3.0], [0.001, 1.0, 0.001]),
    "ground": ([60.0, 40.0, 60.0], [0.001, -1.0, 0.001]),
}


def main():
    import rtamd_loader
    from oracle import pyoracle
    pkg = rtamd_loader.load()
    scene = pkg.default_scene()
    out = {}
    for name, view in VIEWS.items():
        def make(width, height, device):
            t = pkg.RayTracer().create(width, height, device_id=device)
            t.set_samples_per_pixel(16)
            t.set_max_depth(8)
            t.set_aperture(0.0)
            if view is not None:
                t.set_camera(view[0], view[1], 10.0)
            return t
        host = make(240, 135, -1)
        r = pyoracle.render(host.get_params_words(0), scene, want_output=False, want_counters=True)
        c = r["counters"]
        flops = pyoracle.flops_from_counters(c) / c["samples"]
        host.destroy()
        t = make(1920, 1080, 0)
        t.set_frame_overlap(False)
        for _ in range(3):
            t.reset_accumulation()
            t.render_frames(0, 64)
            t.synchronize()
        ms = t.last_kernel_ms()
        t.destroy()
        samples = 1920 * 1080 * 16 * 64
        out[name] = {"flops_per_sample": flops, "segments_per_sample": c["segments"] / c["samples"], "kernel_ms": ms,
                     "samples_per_s": samples / (ms * 1e-3), "tflops": flops * samples / (ms * 1e-3) / 1e12,
                     "of_fp32_valu_peak": flops * samples / (ms * 1e-3) / 157.3e12}
        print(name, json.dumps(out[name]), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "uniform_lanes_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()

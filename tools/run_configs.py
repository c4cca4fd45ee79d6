#!/usr/bin/env python3
"""Times the benchmark configurations C1..C5 on one GPU (kernel time by HIP events, inputs resident).
Usage (GPU box): python tools/run_configs.py [c1 c2 c3 c4 c5 ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (width, height, spp, depth, aperture, frames, scene)
    "c1": (1280, 720, 1, 8, 0.05, 1, "default"),
    "c2": (1920, 1080, 16, 8, 0.0, 64, "default"),
    "c3": (3840, 2160, 64, 16, 0.05, 64, "default"),
    "c4": (7680, 4320, 256, 8, 0.05, 1, "default"),
    "c2f": (1920, 1080, 16, 8, 0.0, 1, "default"),      # one frame per launch, as App::run drives it
    "c2q": (1920, 1080, 4, 12, 0.05, 1, "default"),     # the application's defaults: 4 spp, depth 12
    "c5": (1920, 1080, 16, 50, 0.05, 1, "synthetic4096"),
    "c5s": (480, 270, 16, 50, 0.05, 1, "synthetic4096"),
    "c5m": (960, 540, 16, 50, 0.05, 4, "synthetic4096"),
    "rnd": (1920, 1080, 16, 50, 0.1, 4, "random"),       # the ~480-sphere random scene, camera (13,2,3)
}


def main():
    import rtamd_loader
    pkg = rtamd_loader.load()
    names = sys.argv[1:] or ["c1", "c2", "c3", "c4", "c5s"]
    out = {}
    for name in names:
        w, h, spp, depth, ap, frames, scene = CONFIGS[name]
        t = pkg.RayTracer().create(w, h, device_id=0)
        t.set_samples_per_pixel(spp)
        t.set_max_depth(depth)
        t.set_aperture(ap)
        if scene == "synthetic4096":
            t.set_spheres(pkg.synthetic_scene(4096))
            t.set_camera([3.0, 1.5, 6.0], [-0.4, -0.2, -1.0], 6.0)
        elif scene == "random":
            t.set_spheres(pkg.random_scene())
        t.render_frames(0, frames)       # warm-up (also allocates the per-frame sum image)
        t.synchronize()
        t0 = time.perf_counter()
        t.render_frames(0, frames)
        t.synchronize()
        wall = time.perf_counter() - t0
        launches = max(1, -(-frames * w * h * 16 // (16 << 30))) if frames > 1 else 1
        kernel_ms = sum(t.kernel_ms(i) for i in range(launches))
        samples = w * h * spp * frames
        out[name] = {"samples": samples, "kernel_ms": kernel_ms, "wall_ms": wall * 1e3,
                     "samples_per_s": samples / wall, "trace_launches": launches}
        print(name, json.dumps(out[name]), flush=True)
        t.destroy()
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "configs.json"), "w"), indent=1)


if __name__ == "__main__":
    main()

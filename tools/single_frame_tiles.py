#!/usr/bin/env python3
"""One rt_render call per frame (how the reference's caller drives render(), src/core/App.cpp:403) timed for a few
settings; RT_TILE_LOG2="w h" in the environment overrides the pixel tile of a unit.  Usage (GPU box):
python tools/single_frame_tiles.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import rtamd_loader
    pkg = rtamd_loader.load()
    for name, spp, depth, aperture in (("c2 settings", 16, 8, 0.0), ("application defaults", 4, 12, 0.05), ("1 spp", 1, 8, 0.05),
                                       ("4096 spheres, c5", 16, 50, 0.05), ("4096 spheres, 4 spp", 4, 12, 0.05)):
        t = pkg.RayTracer().create(1920, 1080, device_id=0)
        t.set_samples_per_pixel(spp)
        t.set_max_depth(depth)
        t.set_aperture(aperture)
        if name.startswith("4096"):
            t.set_spheres(pkg.synthetic_scene(4096))
            t.set_camera([3.0, 1.5, 6.0], [-0.4, -0.2, -1.0], 6.0)
        frames = 64
        best = 1e9
        for rep in range(3):
            t.synchronize()
            t0 = time.perf_counter()
            for f in range(frames):
                t.render(f)
            t.synchronize()
            best = min(best, time.perf_counter() - t0)
        print("%-22s tile %-6s %.4f ms per frame  %.3e samples/s" % (name, os.environ.get("RT_TILE_LOG2", "auto"), best / frames * 1e3,
                                                                     1920 * 1080 * spp * frames / best), flush=True)
        t.destroy()


if __name__ == "__main__":
    main()

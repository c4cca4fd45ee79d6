#!/usr/bin/env python3
"""Times scenes beyond the LDS-resident hierarchy (GPU box): the synthetic grid scene at growing sphere
counts, 960x540, 8 spp, depth 16, 4 frames, through the hierarchy; the loop over all spheres for
comparison where it finishes in reasonable time.  python tools/large_scene_timing.py [counts ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import rtamd_loader
    pkg = rtamd_loader.load()
    counts = [int(v) for v in sys.argv[1:]] or [4096, 5000, 16384, 65536, 262144]
    out = {}
    for n in counts:
        scene = pkg.synthetic_scene(n)
        t = pkg.RayTracer().create(960, 540, device_id=0)
        t.set_samples_per_pixel(8)
        t.set_max_depth(16)
        t0 = time.perf_counter()
        t.set_spheres(scene)
        build_ms = (time.perf_counter() - t0) * 1e3
        t.set_camera([3.0, 1.5, 6.0], [-0.4, -0.2, -1.0], 6.0)
        row = {"set_spheres_ms": build_ms, "nodes": int(t.get_scene_hierarchy_links()[1].shape[0])}
        images = {}
        row["tree_groups"] = int(t.get_scene_quadtree()["groups"])
        # hierarchy: the four-way tree of boxes (in LDS or in global memory); depth_first: the tree of balls, one node per trip
        for name, enable in (("hierarchy", True), ("depth_first", 2), ("loop", False)):
            if enable is False and n > 20000:
                continue
            t.set_scene_hierarchy(enable)
            for _ in range(2):
                t.reset_accumulation()
                t.render_frames(0, 4)
                t.synchronize()
            row[name + "_ms"] = t.last_kernel_ms()
            images[name] = t.read_accum()
        row["identical"] = all(bool(np.array_equal(images["hierarchy"].view(np.uint32), im.view(np.uint32))) for im in images.values())
        out[n] = row
        print(n, json.dumps(row), flush=True)
        t.destroy()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "large_scene_timing.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()

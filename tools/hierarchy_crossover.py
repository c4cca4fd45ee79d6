#!/usr/bin/env python3
"""Times the plain sphere loop against the bounding-sphere hierarchy for scenes of growing size
(GPU box): python tools/hierarchy_crossover.py [sizes ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import rtamd_loader
    pkg = rtamd_loader.load()
    sizes = [int(v) for v in sys.argv[1:]] or [32, 48, 64, 96, 128, 256, 512, 1024, 2048, 4096]
    out = {}
    for n in sizes:
        t = pkg.RayTracer().create(960, 540, device_id=0)
        t.set_samples_per_pixel(8)
        t.set_max_depth(16)
        t.set_spheres(pkg.synthetic_scene(n))
        t.set_camera([3.0, 1.5, 6.0], [-0.4, -0.2, -1.0], 6.0)
        row = {}
        images = {}
        for name, enable in (("loop", False), ("hierarchy", True)):
            t.set_scene_hierarchy(enable)
            t.reset_accumulation()
            t.render_frames(0, 4)
            t.synchronize()
            t.reset_accumulation()
            t.render_frames(0, 4)
            t.synchronize()
            row[name + "_ms"] = t.last_kernel_ms()
            images[name] = t.read_accum()
        row["identical"] = bool(np.array_equal(images["loop"].view(np.uint32), images["hierarchy"].view(np.uint32)))
        row["nodes"] = int(t.get_scene_hierarchy()[1].shape[0])
        out[n] = row
        print(n, json.dumps(row), flush=True)
        t.destroy()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "hierarchy_crossover.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()

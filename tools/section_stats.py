#!/usr/bin/env python3
"""Runs the benchmark workload once on the diagnostic build (make -C ray-tracing_amd/csrc stats) and
prints, per code section of the trace kernel, how often a wave executed it and the mean number of
active lanes.  Usage (GPU box):  python tools/section_stats.py [frames]
"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SECTIONS = ["iter", "item", "gen", "disk_it", "norm2", "root", "far_root", "hit", "ris", "ris_it",
            "sky", "lambert", "checker", "metal", "diel", "schlick", "node", "all_primary", "sphere_test"]


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    import rtamd_loader
    pkg = rtamd_loader.load()
    from raytracing_amd import tracer as tr
    path = os.path.join(ROOT, "ray-tracing_amd", "librt_mi355x_stats.so")
    lib = tr.load_library(path)
    tr._lib = lib  # the RayTracer mirror binds to this build
    lib.rt_debug_stats_sections.restype = ctypes.c_int
    lib.rt_debug_read_stats.restype = ctypes.c_int
    lib.rt_debug_read_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    n = lib.rt_debug_stats_sections()
    assert n == len(SECTIONS), (n, len(SECTIONS))
    big = os.environ.get("RT_STATS_SCENE") == "synthetic4096"
    width, height = (960, 540) if big else (1920, 1080)
    if os.environ.get("RT_STATS_SIZE"):
        width, height = [int(v) for v in os.environ["RT_STATS_SIZE"].split()]
    t = pkg.RayTracer().create(width, height, device_id=0)
    t.set_samples_per_pixel(16)
    t.set_max_depth(50 if big else 8)
    t.set_aperture(0.05 if big else 0.0)
    if big:
        t.set_spheres(pkg.synthetic_scene(4096))
        t.set_camera([3.0, 1.5, 6.0], [-0.4, -0.2, -1.0], 6.0)
    lib.rt_debug_stats_phases.restype = ctypes.c_int
    nph = lib.rt_debug_stats_phases()
    buf = (ctypes.c_uint64 * (2 * n + nph))()
    assert lib.rt_debug_read_stats(t.handle, buf, 1) == 0   # allocates + zeroes the counters
    t.render_frames(0, frames)
    t.synchronize()
    assert lib.rt_debug_read_stats(t.handle, buf, 0) == 0
    samples = width * height * 16 * frames
    iters = buf[0]
    out = {}
    print("%-10s %14s %10s %12s %10s" % ("section", "wave execs", "per iter", "mean lanes", "lanes/64"))
    for i, name in enumerate(SECTIONS):
        ex, ln = buf[2 * i], buf[2 * i + 1]
        mean = ln / ex if ex else 0.0
        out[name] = {"wave_execs": int(ex), "lane_execs": int(ln)}
        print("%-10s %14d %10.4f %12.2f %10.3f" % (name, ex, ex / iters if iters else 0, mean, mean / 64.0))
    phases = ["norm2", "intersect", "hit", "item+jitter", "sampler", "norm1", "shade", "tail(write+camera)"]
    cyc = [buf[2 * n + i] for i in range(nph)]
    tot = float(sum(cyc))
    print("phase shares of one wave's timeline (shader-clock cycles between stamps; a stamp costs ~40 cycles):")
    for name, c in zip(phases, cyc):
        print("  %-20s %6.1f %%   %8.1f cycles/iteration" % (name, 100.0 * c / tot, c / iters))
    out["_phases"] = dict(zip(phases, [int(c) for c in cyc]))
    print("wave iterations per wave-sample: %.3f (ideal = segments per sample: 1.946 for the default scene)" % (iters / (samples / 64.0)))
    if big and os.environ.get("RT_TREE_DEPTH_FIRST"):
        # every visit of a hierarchy node reads 32 bytes of LDS: the two possible successors are fetched while the node is tested
        visits = out["node"]["lane_execs"] / samples
        print("hierarchy: %.1f node visits per sample, %.0f bytes of LDS per sample at 32 bytes per visit" % (visits, 32.0 * visits))
        out["_hierarchy"] = {"node_visits_per_sample": visits, "lds_bytes_per_sample": 32.0 * visits}
    elif big:
        # the four-way tree: a box group is 96 bytes of LDS, a group of four spheres 64
        boxes = out["node"]["lane_execs"] / samples
        spheres = out["sphere_test"]["lane_execs"] / samples
        lds = 96.0 * boxes + 64.0 * spheres
        print("hierarchy: %.1f box groups and %.1f groups of spheres visited per sample, %.0f bytes of LDS per sample "
              "(96 and 64 bytes per visit)" % (boxes, spheres, lds))
        out["_hierarchy"] = {"box_groups_per_sample": boxes, "sphere_groups_per_sample": spheres, "lds_bytes_per_sample": lds}
    json.dump({"frames": frames, "samples": samples, "sections": out}, open(os.path.join(ROOT, "gpurun_out", "section_stats.json"), "w"), indent=1)
    t.destroy()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Load balance of the two row splits, measured on one GPU: renders a workload as G contiguous row bands
and as G interleaved stripe shards (what G GPUs would each render) and prints the trace-kernel time of
every shard; the slowest shard is what a G-GPU step would wait for.
Usage (GPU box): python tools/shard_balance.py [c2|c4] [G] [stripe_rows]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import rtamd_loader
    pkg = rtamd_loader.load()
    from raytracing_amd import dist as rtdist
    import bench
    name = sys.argv[1] if len(sys.argv) > 1 else "c2"
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    wl = bench.WORKLOADS[name]
    stripe_rows = int(sys.argv[3]) if len(sys.argv) > 3 else (16 if wl["height"] >= 4000 else 4)
    t = pkg.RayTracer().create(wl["width"], wl["height"], device_id=0)
    t.set_samples_per_pixel(wl["spp"])
    t.set_max_depth(wl["depth"])
    t.set_aperture(wl["aperture"])
    frames = wl["frames"]
    t.set_frame_overlap(False)   # shard launches are timed one by one here
    t.render_frames(0, frames)
    t.synchronize()
    t.render_frames(0, frames)
    whole = t.kernel_ms(0)
    out = {"workload": name, "ranks": world, "stripe_rows": stripe_rows, "whole_image_ms": whole}
    import time
    for mode in ("bands", "stripes"):
        ms = []
        for rep in range(3):
            # the shards are queued back to back (no host synchronisation in between, so the GPU does not
            # idle and clock down between them) and their kernel times are read from the event ring afterwards
            t.synchronize()
            t0 = time.perf_counter()
            for r in range(world):
                if mode == "bands":
                    b, e = rtdist.row_bands(wl["height"], world)[r]
                    t.render_shard(0, frames, 1, b, e, write_output=False)
                else:
                    t.render_stripes(0, frames, 1, stripe_rows, r, world, write_output=False)
            t.synchronize()
            wall = (time.perf_counter() - t0) * 1e3
            ms = [t.kernel_ms(world - 1 - r) for r in range(world)]
        out[mode + "_wall_ms_all_shards_back_to_back"] = wall
        out[mode] = {"shard_ms": [round(m, 4) for m in ms], "sum_ms": sum(ms), "max_ms": max(ms),
                     "speedup_if_parallel": whole / max(ms), "imbalance_max_over_mean": max(ms) / (sum(ms) / world)}
    # the same shards queued back to back with the library's launch overlap on: what a rank's consecutive steps cost
    t.set_frame_overlap(True)
    for rep in range(3):
        t.synchronize()
        t0 = time.perf_counter()
        for r in range(world):
            t.render_stripes(0, frames, 1, stripe_rows, r, world, write_output=False)
        t.synchronize()
        out["stripes_wall_ms_all_shards_overlapped"] = (time.perf_counter() - t0) * 1e3
    print(json.dumps(out, indent=1))
    t.destroy()


if __name__ == "__main__":
    main()

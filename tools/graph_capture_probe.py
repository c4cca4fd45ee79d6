#!/usr/bin/env python3
"""Captures rt_render_frames_static (the entry point that allocates nothing) into a HIP graph and replays it.

A stream capture forbids allocation and synchronisation, which is what that entry point is for.  The probe
captures one call on a PyTorch stream, replays the graph twice and compares the accumulator and the rgba8 image
with an ordinary rt_render_frames call of the same frames, bit for bit.  Usage (GPU box):
python tools/graph_capture_probe.py [WIDTH HEIGHT FRAMES]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import rtamd_loader
    pkg = rtamd_loader.load()
    w, h, frames = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (640, 360, 4)

    plain = pkg.RayTracer().create(w, h, device_id=0)
    plain.set_samples_per_pixel(2)
    plain.set_max_depth(6)
    plain.render_frames(0, frames)
    want_accum, want_output = plain.read_accum(), plain.read_output()

    stream = torch.cuda.Stream()
    t = pkg.RayTracer().create(w, h, device_id=0)
    t.set_samples_per_pixel(2)
    t.set_max_depth(6)
    t.set_stream(stream.cuda_stream)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream):
        t.render_frames_static(0, frames)
    print("captured", flush=True)
    for replay in range(2):
        graph.replay()
        torch.cuda.synchronize()
        accum, output = t.read_accum(), t.read_output()
        same = np.array_equal(accum.view(np.uint32), want_accum.view(np.uint32)) and np.array_equal(output, want_output)
        print("replay %d: %s" % (replay, "bit-identical to rt_render_frames" if same else "MISMATCH"), flush=True)
        if not same:
            sys.exit(1)
    # the handle is still usable outside the graph, scene changes included
    t.set_spheres(pkg.default_scene())
    t.render_frames(0, frames)
    ok = np.array_equal(t.read_accum().view(np.uint32), want_accum.view(np.uint32))
    print("after the capture: %s" % ("ok" if ok else "MISMATCH"), flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()

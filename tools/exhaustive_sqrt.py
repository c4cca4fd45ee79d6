#!/usr/bin/env python3
"""Exhaustive check of the kernel's short square root against sqrtf on the device for every binary32 x
in [2^-90, 2^100) -- the range the short path serves (GPU box).  python tools/exhaustive_sqrt.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import rtamd_loader
    rtamd_loader.load()
    from raytracing_amd import tracer
    lo = np.float32(2.0 ** -90).view(np.uint32).item()
    hi = np.float32(2.0 ** 100).view(np.uint32).item() - 1
    chunk = 1 << 24
    bad = 0
    total = 0
    for n, start in enumerate(range(lo, hi + 1, chunk)):
        bits = np.arange(start, min(start + chunk, hi + 1), dtype=np.uint32)
        cheap, full = tracer.probe_arith(1, bits.view(np.float32))
        mismatch = cheap.view(np.uint32) != full.view(np.uint32)
        bad += int(mismatch.sum())
        total += bits.size
        if mismatch.any() and bad < 50:
            i = int(np.nonzero(mismatch)[0][0])
            print("mismatch at x = %r: %r vs %r" % (bits.view(np.float32)[i], cheap[i], full[i]), flush=True)
        if n % 8 == 0:
            print("checked %d values, %d mismatches" % (total, bad), flush=True)
    print("done: %d values, %d mismatches" % (total, bad), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

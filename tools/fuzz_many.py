#!/usr/bin/env python3
"""A long run of the randomised parity cases of tests/test_fuzz_gpu.py (HIP path against the CPU oracle, bit for
bit) over a range of seeds.  Usage (GPU box): python tools/fuzz_many.py FIRST_SEED COUNT [big]
(big: scenes of 401 to 9000 spheres on small images)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    big = len(sys.argv) > 3 and sys.argv[3] == "big"
    import rtamd_loader
    from oracle import pyoracle
    import test_fuzz_gpu as fz
    pkg = rtamd_loader.load()
    t0 = time.time()
    patterns = {}
    for i in range(count):
        seed = first + i
        c = fz._case(seed, specials=(seed % 3 == 0), big=big)
        patterns[c["pattern"]] = patterns.get(c["pattern"], 0) + 1
        try:
            fz._run_case(pkg, pyoracle, c)
        except BaseException as e:
            print("MISMATCH at seed %d: %s" % (seed, str(e)[:300]), flush=True)
            sys.exit(1)
        if i % (20 if big else 200) == (19 if big else 199):
            print("%d cases ok (%.0f s) %s" % (i + 1, time.time() - t0, patterns), flush=True)
    print("all %d cases ok: seeds %d..%d, patterns %s" % (count, first, first + count - 1, patterns), flush=True)


if __name__ == "__main__":
    main()

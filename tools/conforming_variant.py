#!/usr/bin/env python3
"""Strict oracle against the conforming-variant build of the same restatement (oracle/variant_study.py):
diverged random streams and RMSE of the converged radiance.  CPU only.
Usage: python tools/conforming_variant.py [width height] > profiles/<round>/conforming_variant.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import variant_study  # noqa: E402

if __name__ == "__main__":
    w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 36)
    print(json.dumps(variant_study.study(width=w, height=h), indent=1))

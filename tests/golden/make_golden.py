#!/usr/bin/env python3
"""Generates tests/golden/host_params.json, tests/golden/rng_kat.json, tests/golden/glm_builtins.json and
tests/golden/caller_camera.json.

host_params.json: GPUSphere / GPUParams bytes printed by oracle/_ref/ref_params_probe, a probe
linked against the reference's own src/rt/RayTracer.cpp (built by `make -C oracle ref`, only
possible where /root/reference is mounted).  rng_kat.json: known answers of the shader's integer
RNG (shaders/raytrace.comp.glsl:37-55) computed here with Python integers, independently of the
C oracle and of the HIP code.  glm_builtins.json: the GLSL built-ins the shader calls (dot,
normalize, inversesqrt, reflect, mix, mod, floor, min, max, length, cross, pow) evaluated by
oracle/_ref/ref_glm_probe with the reference's vendored GLM 1.0.1 headers; inputs and results as
hex words of binary32 values.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
PROBE = os.path.join(ROOT, "oracle", "_ref", "ref_params_probe")
GLM_PROBE = os.path.join(ROOT, "oracle", "_ref", "ref_glm_probe")
CAMERA_PROBE = os.path.join(ROOT, "oracle", "_ref", "ref_camera_probe")

# name -> probe argv.  "default": create() + setSamplesPerPixel(4) + setAperture(a) as App::run does
# (src/core/App.cpp:153-156); otherwise: width height aperture vfov focus pos dir through the setters.
CASES = {
    "default_1280x720_ap0.05": ["default", "1280", "720", "0.05"],
    "default_1920x1080_ap0.05": ["default", "1920", "1080", "0.05"],
    "default_1920x1080_ap0": ["default", "1920", "1080", "0"],
    "default_3840x2160_ap0.05": ["default", "3840", "2160", "0.05"],
    "default_7680x4320_ap0.05": ["default", "7680", "4320", "0.05"],
    "default_333x77_ap0.2": ["default", "333", "77", "0.2"],
    "custom_640x360": ["640", "360", "0.1", "45", "7.5", "3", "4", "5", "-0.3", "-0.4", "-0.5"],
    "custom_fovclamp_nofocus": ["512", "512", "-1", "150", "-1", "0", "2", "10", "0", "0", "-2"],
    "custom_lowfov": ["800", "600", "0.02", "1", "0.25", "-7", "0.5", "-2", "1", "0.1", "0.3"],
}


def run_probe(args):
    out = subprocess.check_output([PROBE] + args, text=True)
    rec = {"argv": args}
    for line in out.splitlines():
        parts = line.split()
        if parts[0] == "sizeof_sphere":
            rec["sizeof_sphere"] = int(parts[1])
            rec["sizeof_params"] = int(parts[3])
            rec["nspheres"] = int(parts[5])
        else:
            rec[parts[0]] = parts[1:]
    return rec


def hash1(v):
    v = ((v ^ (v >> 17)) * 0xED5AD4BB) & 0xFFFFFFFF
    v = ((v ^ (v >> 11)) * 0xAC4C1B51) & 0xFFFFFFFF
    v = ((v ^ (v >> 15)) * 0x31848BAB) & 0xFFFFFFFF
    return v


def hash3(x, y, z):
    return hash1(x) ^ hash1(y) ^ hash1(z)


def lcg(s):
    return (1664525 * s + 1013904223) & 0xFFFFFFFF


def rng_kat():
    seeds = [(0, 0, 0), (1, 2, 3), (960, 540, 0), (1919, 1079, 63), (5, 7, 0), (7, 5, 0),
             (7679, 4319, 255), (0, 0, 1), (4294967295, 0, 0), (123456, 654321, 4096)]
    out = []
    for (x, y, z) in seeds:
        h = hash3(x, y, z)
        s = h
        states = []
        for _ in range(8):
            s = lcg(s)
            states.append(s)
        # rand = float(state & 0xFFFFFF) / 2^24 is exact in binary32: store the 24-bit numerator
        out.append({"xyz": [x, y, z], "hash": h, "lcg": states, "rand_num24": [t & 0xFFFFFF for t in states]})
    return out


def glm_builtins():
    out = []
    for line in subprocess.check_output([GLM_PROBE], text=True).splitlines():
        left, right = line.split("->")
        parts = left.split()
        out.append({"op": parts[0], "in": parts[1:], "out": right.split()})
    return out


def caller_camera():
    """The camera arithmetic of the reference's frame loop (src/core/App.cpp:180-183, :299-362) on a scripted
    event sequence, evaluated by oracle/_ref/ref_camera_probe with the reference's vendored GLM: per step the
    events and the resulting camPos, camDir, yaw, pitch as hex words."""
    out = {"init": None, "steps": []}
    for line in subprocess.check_output([CAMERA_PROBE], text=True).splitlines():
        parts = line.split()
        if parts[0] == "init":
            out["init"] = parts[1:]
        else:
            out["steps"].append({"dx": float(parts[1]), "dy": float(parts[2]), "keys": "" if parts[3] == "-" else parts[3],
                                 "dt": float(parts[4]), "state": parts[6:]})
    return out


def main():
    if not os.path.exists(PROBE) or not os.path.exists(GLM_PROBE):
        sys.exit("build oracle/_ref first: make -C oracle ref (needs /root/reference)")
    params = {name: run_probe(args) for name, args in CASES.items()}
    with open(os.path.join(HERE, "host_params.json"), "w") as f:
        json.dump(params, f, indent=1, sort_keys=True)
    with open(os.path.join(HERE, "rng_kat.json"), "w") as f:
        json.dump(rng_kat(), f, indent=1)
    vectors = glm_builtins()
    with open(os.path.join(HERE, "glm_builtins.json"), "w") as f:
        json.dump(vectors, f, separators=(",", ":"))
    if os.path.exists(CAMERA_PROBE):
        with open(os.path.join(HERE, "caller_camera.json"), "w") as f:
            json.dump(caller_camera(), f, indent=1)
    print("wrote host_params.json (%d cases), rng_kat.json and glm_builtins.json (%d vectors)" % (len(params), len(vectors)))


if __name__ == "__main__":
    main()

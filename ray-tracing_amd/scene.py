"""Sphere scenes in the reference's GPU layout (GPUSphere, src/rt/RayTracer.h:12-17).

A scene is a float32 array of shape (N, 12): centre.xyz, radius | albedo.rgb, 0 |
material (0 lambert, 1 metal, 2 dielectric), fuzz, refIdx, flags (>0.5 = checker).

Scenes can be kept in two file forms: `.json` (a list of named fields per sphere, human-editable) and
`.spheres` (the raw little-endian float32 records, 48 bytes each, exactly what rt_set_spheres takes).
"""
import json
import os

import numpy as np

LAMBERT, METAL, DIELECTRIC = 0.0, 1.0, 2.0


def default_scene():
    """The five spheres RayTracer::buildScene hard-codes (src/rt/RayTracer.cpp:105-138)."""
    return np.array([
        [0.0, -1000.0, 0.0, 1000.0, 0.75, 0.8, 0.9, 0.0, LAMBERT, 0.0, 1.0, 1.0],
        [0.0, 1.0, 0.0, 1.0, 0.9, 0.25, 0.25, 0.0, LAMBERT, 0.0, 1.0, 0.0],
        [-4.0, 1.0, 0.0, 1.0, 1.0, 1.0, 1.0, 0.0, DIELECTRIC, 0.0, 1.5, 0.0],
        [4.0, 1.0, 0.0, 1.0, 0.95, 0.65, 0.15, 0.0, METAL, 0.03, 1.0, 0.0],
        [2.5, 0.5, 2.5, 0.5, 0.95, 0.95, 0.98, 0.0, METAL, 0.0, 1.0, 0.0],
    ], dtype=np.float32)


class _Lcg:
    """The shader's LCG (raytrace.comp.glsl:46-55) reused as the scene generator's RNG."""

    def __init__(self, seed):
        self.state = seed & 0xFFFFFFFF

    def rand(self):
        self.state = (1664525 * self.state + 1013904223) & 0xFFFFFFFF
        return np.float32(self.state & 0x00FFFFFF) / np.float32(16777216.0)


def synthetic_scene(count=4096, seed=0x5EED):
    """Deterministic mixed-material scene: a checker ground sphere plus count-1 small spheres
    (radius 0.2) on a jittered square grid; about 80 % lambert, 15 % metal (fuzz in [0, 0.5)),
    5 % glass (refIdx 1.5).  The same bytes come out on every machine."""
    if count < 1:
        raise ValueError("count must be >= 1")
    rng = _Lcg(seed)
    out = np.zeros((count, 12), dtype=np.float32)
    out[0] = [0.0, -1000.0, 0.0, 1000.0, 0.5, 0.5, 0.5, 0.0, LAMBERT, 0.0, 1.0, 1.0]
    side = int(np.ceil(np.sqrt(count)))
    spacing = np.float32(0.5)
    half = np.float32(side) * spacing * np.float32(0.5)
    for i in range(1, count):
        gx, gz = (i - 1) % side, (i - 1) // side
        jx = (rng.rand() - np.float32(0.5)) * np.float32(0.1)
        jz = (rng.rand() - np.float32(0.5)) * np.float32(0.1)
        cx = np.float32(gx) * spacing - half + np.float32(0.25) + jx
        cz = np.float32(gz) * spacing - half + np.float32(0.25) + jz
        choose = rng.rand()
        r, g, b = rng.rand(), rng.rand(), rng.rand()
        fuzz = rng.rand() * np.float32(0.5)
        if choose < np.float32(0.8):
            out[i] = [cx, 0.2, cz, 0.2, r * r, g * g, b * b, 0.0, LAMBERT, 0.0, 1.0, 0.0]
        elif choose < np.float32(0.95):
            out[i] = [cx, 0.2, cz, 0.2, 0.5 + 0.5 * r, 0.5 + 0.5 * g, 0.5 + 0.5 * b, 0.0, METAL, fuzz, 1.0, 0.0]
        else:
            out[i] = [cx, 0.2, cz, 0.2, 1.0, 1.0, 1.0, 0.0, DIELECTRIC, 0.0, 1.5, 0.0]
    return out


def random_scene(seed=0x1234):
    """The classic random scene of small spheres around three large ones (22 x 22 grid), generated with
    the shader's LCG so that the same bytes come out everywhere: up to 488 spheres."""
    rng = _Lcg(seed)
    out = [[0.0, -1000.0, 0.0, 1000.0, 0.5, 0.5, 0.5, 0.0, LAMBERT, 0.0, 1.0, 1.0]]
    for a in range(-11, 11):
        for b in range(-11, 11):
            choose = rng.rand()
            cx = np.float32(a) + np.float32(0.9) * rng.rand()
            cz = np.float32(b) + np.float32(0.9) * rng.rand()
            if (cx - np.float32(4.0)) ** 2 + cz ** 2 <= np.float32(0.81):
                continue
            r, g, bl, fz = rng.rand(), rng.rand(), rng.rand(), rng.rand()
            if choose < np.float32(0.8):
                out.append([cx, 0.2, cz, 0.2, r * r, g * g, bl * bl, 0.0, LAMBERT, 0.0, 1.0, 0.0])
            elif choose < np.float32(0.95):
                out.append([cx, 0.2, cz, 0.2, 0.5 + 0.5 * r, 0.5 + 0.5 * g, 0.5 + 0.5 * bl, 0.0, METAL, 0.5 * fz, 1.0, 0.0])
            else:
                out.append([cx, 0.2, cz, 0.2, 1.0, 1.0, 1.0, 0.0, DIELECTRIC, 0.0, 1.5, 0.0])
    out.append([0.0, 1.0, 0.0, 1.0, 1.0, 1.0, 1.0, 0.0, DIELECTRIC, 0.0, 1.5, 0.0])
    out.append([-4.0, 1.0, 0.0, 1.0, 0.4, 0.2, 0.1, 0.0, LAMBERT, 0.0, 1.0, 0.0])
    out.append([4.0, 1.0, 0.0, 1.0, 0.7, 0.6, 0.5, 0.0, METAL, 0.0, 1.0, 0.0])
    return np.array(out, dtype=np.float32)


_MATERIAL_NAMES = {0: "lambert", 1: "metal", 2: "dielectric"}


def save_scene(path, spheres):
    """Writes a scene as .json (named fields) or, for any other suffix, as raw float32 records."""
    arr = np.ascontiguousarray(spheres, dtype=np.float32).reshape(-1, 12)
    if str(path).endswith(".json"):
        items = []
        for s in arr:
            items.append({"center": [float(v) for v in s[0:3]], "radius": float(s[3]),
                          "albedo": [float(v) for v in s[4:7]],
                          "material": _MATERIAL_NAMES.get(int(s[8] + 0.5), "dielectric"),
                          "fuzz": float(s[9]), "ref_idx": float(s[10]), "checker": bool(s[11] > 0.5)})
        with open(path, "w") as f:
            json.dump({"spheres": items}, f, indent=1)
    else:
        arr.astype("<f4").tofile(path)


def load_scene(path):
    """Reads a scene written by save_scene (or hand-written JSON in the same shape)."""
    if str(path).endswith(".json"):
        with open(path) as f:
            doc = json.load(f)
        names = {v: k for k, v in _MATERIAL_NAMES.items()}
        out = []
        for s in doc["spheres"]:
            mat = s.get("material", "lambert")
            if mat not in names:
                raise ValueError("unknown material %r" % (mat,))
            out.append(list(s["center"]) + [s["radius"]] + list(s.get("albedo", [1.0, 1.0, 1.0])) + [0.0] +
                       [float(names[mat]), s.get("fuzz", 0.0), s.get("ref_idx", 1.0), 1.0 if s.get("checker") else 0.0])
        arr = np.array(out, dtype=np.float32).reshape(-1, 12)
    else:
        size = os.path.getsize(path)
        if size == 0 or size % 48 != 0:
            raise ValueError("%s does not hold whole 48-byte sphere records" % (path,))
        raw = np.fromfile(path, dtype="<f4")
        arr = raw.reshape(-1, 12).astype(np.float32)
    if arr.shape[0] == 0:
        raise ValueError("scene is empty")
    return arr

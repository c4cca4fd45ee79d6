"""Sphere scenes in the reference's GPU layout (GPUSphere, src/rt/RayTracer.h:12-17).

A scene is a float32 array of shape (N, 12): centre.xyz, radius | albedo.rgb, 0 |
material (0 lambert, 1 metal, 2 dielectric), fuzz, refIdx, flags (>0.5 = checker).
"""
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

"""MI355X-native sphere path tracer: the traced path of RippleDomain/Ray-Tracing behind a C ABI.

The directory name carries a hyphen, so import it through rtamd_loader.load()
(repo root), which registers it as the module `raytracing_amd`.
"""
from .tracer import (ABI, GPUParams, GPUSphere, LIB_PATH, RayTracer, load_library, probe_arith, probe_hit_sphere,
                     probe_pow, probe_rng)
from .scene import default_scene, load_scene, random_scene, save_scene, synthetic_scene
from .caller import FrameLoop
from . import image

__all__ = ["ABI", "GPUParams", "GPUSphere", "LIB_PATH", "RayTracer", "load_library", "probe_hit_sphere",
           "probe_pow", "probe_rng", "probe_arith", "default_scene", "synthetic_scene", "random_scene", "load_scene", "save_scene", "FrameLoop", "image"]

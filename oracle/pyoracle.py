"""ctypes binding of the CPU oracle (oracle/librt_oracle.so) -- TEST INFRASTRUCTURE ONLY.

Imported by tests/, by __graft_entry__.smoke() and by the cpu_baseline leg of
bench.py as the checker.  The shipped package never imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librt_oracle.so")


class Sphere(ctypes.Structure):
    _fields_ = [("center_radius", ctypes.c_float * 4),
                ("albedo", ctypes.c_float * 4),
                ("misc", ctypes.c_float * 4)]


class Params(ctypes.Structure):
    _fields_ = [("origin_lens", ctypes.c_float * 4),
                ("lower_left", ctypes.c_float * 4),
                ("horizontal", ctypes.c_float * 4),
                ("vertical", ctypes.c_float * 4),
                ("u", ctypes.c_float * 4),
                ("v", ctypes.c_float * 4),
                ("w", ctypes.c_float * 4),
                ("frame_sample_depth_count", ctypes.c_uint32 * 4),
                ("resolution", ctypes.c_float * 4),
                ("inv_resolution", ctypes.c_float * 4)]


COUNTER_FIELDS = ["samples", "disk_iters", "segments", "sphere_tests", "disc_nonneg", "far_roots",
                  "accepts", "sky", "lambert", "lambert_checker", "metal", "metal_absorbed",
                  "dielectric", "diel_schlick", "diel_refract", "sphere_iters", "depth_exhausted",
                  "rand_draws", "pixels", "diel_front"]


class Counters(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in COUNTER_FIELDS]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in COUNTER_FIELDS}


def build(force=False):
    """Compile the oracle's C restatement (and oracle/_ref when the reference tree is present)."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "rt_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "oracle"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src/rt"):
        subprocess.call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


_lib = None
_variant = None
_VARIANT_PATH = os.path.join(_HERE, "librt_oracle_variant.so")


def cpu_has_fma():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return " fma " in line + " "
    except OSError:
        pass
    return False


def variant_lib():
    """The same restatement built as another conforming GLSL implementation would evaluate it (contracted
    multiply-adds, the C library's pow): oracle/Makefile, target `variant`.  Only for measuring how far such an
    implementation lands from the strict oracle; never a reference for anything.  None on a CPU without FMA."""
    global _variant
    if _variant is None:
        if not cpu_has_fma():
            return None
        src = os.path.join(_HERE, "rt_oracle.c")
        if not os.path.exists(_VARIANT_PATH) or os.path.getmtime(_VARIANT_PATH) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "variant"], stdout=subprocess.DEVNULL)
        _variant = _declare(ctypes.CDLL(_VARIANT_PATH))
    return _variant


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = _declare(ctypes.CDLL(_LIB_PATH))
    return _lib


def _declare(L):
    u32p = ctypes.POINTER(ctypes.c_uint32)
    f32p = ctypes.POINTER(ctypes.c_float)
    L.rto_hash3.restype = ctypes.c_uint32
    L.rto_hash3.argtypes = [ctypes.c_uint32] * 3
    L.rto_lcg.restype = ctypes.c_uint32
    L.rto_lcg.argtypes = [u32p]
    L.rto_rand.restype = ctypes.c_float
    L.rto_rand.argtypes = [u32p]
    for fn in (L.rto_log2f, L.rto_exp2f):
        fn.restype = ctypes.c_float
        fn.argtypes = [ctypes.c_float]
    L.rto_powf.restype = ctypes.c_float
    L.rto_powf.argtypes = [ctypes.c_float, ctypes.c_float]
    L.rto_default_scene.argtypes = [ctypes.POINTER(Sphere)]
    L.rto_normalize3.argtypes = [f32p, f32p]
    L.rto_builtin.restype = ctypes.c_int
    L.rto_builtin.argtypes = [ctypes.c_char_p, f32p, f32p]
    L.rto_make_camera_params.argtypes = [f32p, f32p, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                         ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(Params)]
    L.rto_default_aim.argtypes = [f32p, f32p, f32p]
    L.rto_render.restype = ctypes.c_int
    L.rto_render.argtypes = [ctypes.POINTER(Params), ctypes.POINTER(Sphere), ctypes.c_void_p, ctypes.c_void_p,
                             ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int,
                             ctypes.POINTER(Counters)]
    L.rto_resolve.restype = None
    L.rto_resolve.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32]
    L.rto_hit_sphere.restype = ctypes.c_int
    L.rto_hit_sphere.argtypes = [ctypes.POINTER(Sphere), f32p, f32p, ctypes.c_float, ctypes.c_float,
                                 f32p, f32p, f32p, ctypes.POINTER(ctypes.c_int)]
    L.rto_sample.argtypes = [ctypes.POINTER(Params), ctypes.POINTER(Sphere), ctypes.c_uint32, ctypes.c_uint32,
                             u32p, f32p]
    return L


def _f3(v):
    return (ctypes.c_float * 3)(*[float(x) for x in v])


def hash3(x, y, z):
    return int(lib().rto_hash3(x & 0xFFFFFFFF, y & 0xFFFFFFFF, z & 0xFFFFFFFF))


def lcg_sequence(seed, n):
    st = ctypes.c_uint32(seed)
    return [int(lib().rto_lcg(ctypes.byref(st))) for _ in range(n)]


def rand_sequence(seed, n):
    st = ctypes.c_uint32(seed)
    return [float(lib().rto_rand(ctypes.byref(st))) for _ in range(n)]


def default_scene():
    """The five spheres of src/rt/RayTracer.cpp:105-138 as a (5, 12) float32 array."""
    arr = (Sphere * 5)()
    lib().rto_default_scene(arr)
    return np.frombuffer(bytes(arr), dtype=np.float32).reshape(5, 12).copy()


def normalize3(v):
    out = (ctypes.c_float * 3)()
    lib().rto_normalize3(_f3(v), out)
    return [out[0], out[1], out[2]]


def builtin(op, inputs):
    """The oracle's GLSL built-in `op` on float32 inputs -> list of float32 results (see rto_builtin)."""
    arr = np.ascontiguousarray(inputs, dtype=np.float32)
    buf = (ctypes.c_float * 8)(*([0.0] * 8))
    for i, v in enumerate(arr):
        buf[i] = v
    out = (ctypes.c_float * 4)()
    n = lib().rto_builtin(op.encode(), buf, out)
    if n < 0:
        raise ValueError("unknown built-in %r" % (op,))
    return np.array(out[:n], dtype=np.float32)


def default_aim(pos=(13.0, 2.0, 3.0)):
    d = (ctypes.c_float * 3)()
    f = ctypes.c_float()
    lib().rto_default_aim(_f3(pos), d, ctypes.byref(f))
    return [d[0], d[1], d[2]], f.value


def make_params(pos, stored_dir, aperture, vfov, focus, width, height,
                frame=0, spp=1, max_depth=1, sphere_count=0):
    """GPUParams (160 bytes) as a numpy uint32[40] view of the struct."""
    p = Params()
    lib().rto_make_camera_params(_f3(pos), _f3(stored_dir), aperture, vfov, focus, width, height, ctypes.byref(p))
    p.frame_sample_depth_count[0] = frame
    p.frame_sample_depth_count[1] = spp
    p.frame_sample_depth_count[2] = max_depth
    p.frame_sample_depth_count[3] = sphere_count
    return p


def params_words(p):
    return np.frombuffer(bytes(p), dtype=np.uint32).copy()


def params_from_words(words):
    w = np.ascontiguousarray(words, dtype=np.uint32)
    assert w.size == 40
    return Params.from_buffer_copy(w.tobytes())


def render(params, spheres, accum=None, want_output=True, want_rng=False, y0=0, y1=None, nthreads=None,
           want_counters=False, library=None):
    """One dispatch of the shader over rows [y0, y1).  spheres: (N, 12) float32.  library: the build to run
    (default: the strict oracle; variant_lib() for the conforming-variant build).

    Returns dict(accum=(H,W,4) float32, output=(H,W,4) uint8 or None, rng=(H,W) uint32 or None,
    counters=dict or None).  accum is updated in place when given.
    """
    if isinstance(params, np.ndarray):
        params = params_from_words(params)
    W = int(params.resolution[0])
    H = int(params.resolution[1])
    if y1 is None:
        y1 = H
    if nthreads is None:
        nthreads = min(os.cpu_count() or 1, 32)
    sph = np.ascontiguousarray(spheres, dtype=np.float32).reshape(-1, 12)
    assert sph.shape[0] >= params.frame_sample_depth_count[3]
    if accum is None:
        accum = np.zeros((H, W, 4), dtype=np.float32)
    assert accum.dtype == np.float32 and accum.shape == (H, W, 4) and accum.flags["C_CONTIGUOUS"]
    out = np.zeros((H, W, 4), dtype=np.uint8) if want_output else None
    rng = np.zeros((H, W), dtype=np.uint32) if want_rng else None
    cnt = Counters() if want_counters else None
    rc = (library or lib()).rto_render(ctypes.byref(params), sph.ctypes.data_as(ctypes.POINTER(Sphere)),
                          accum.ctypes.data, out.ctypes.data if out is not None else None,
                          rng.ctypes.data if rng is not None else None, y0, y1, nthreads,
                          ctypes.byref(cnt) if cnt is not None else None)
    assert rc == 0
    return {"accum": accum, "output": out, "rng": rng, "counters": cnt.as_dict() if cnt else None}


def resolve(accum, sample_count):
    """The display tail alone (raytrace.comp.glsl:337-342): accum (..., 4) float32 -> (..., 4) uint8."""
    a = np.ascontiguousarray(accum, dtype=np.float32)
    assert a.shape[-1] == 4
    out = np.zeros(a.shape, dtype=np.uint8)
    lib().rto_resolve(a.ctypes.data, out.ctypes.data, a.size // 4, int(sample_count) & 0xFFFFFFFF)
    return out


def hit_sphere(sphere12, o, d, tmin, tmax):
    s = Sphere.from_buffer_copy(np.ascontiguousarray(sphere12, dtype=np.float32).tobytes())
    t = ctypes.c_float()
    p = (ctypes.c_float * 3)()
    n = (ctypes.c_float * 3)()
    front = ctypes.c_int()
    hit = lib().rto_hit_sphere(ctypes.byref(s), _f3(o), _f3(d), tmin, tmax, ctypes.byref(t), p, n, ctypes.byref(front))
    if not hit:
        return None
    return {"t": t.value, "p": list(p), "n": list(n), "front": bool(front.value)}


def flops_from_counters(c):
    """Algorithmic FLOP total from the oracle's event counters.

    Convention: add/sub/mul/div/sqrt/inversesqrt/floor/pow count 1 each; compares, selects,
    negations, min/max, integer work and int<->float conversions count 0; a fused multiply-add
    gets no extra credit; operations are counted as written in raytrace.comp.glsl (vector
    operations count one per component actually written in the source, `a = dot(d,d)` is
    counted in every hitSphere call).
    """
    f = 0
    f += c["rand_draws"] * 1           # :54  float(..) / float(2^24)
    # :317-330 per camera sample: uCoord 2, vCoord 4, lens*disk 3, offset 9, dir 18, origin 3,
    # normalize 9, accum += 3
    f += c["samples"] * (2 + 4 + 3 + 9 + 18 + 3 + 9 + 3)
    f += c["disk_iters"] * (6 + 3)     # :81-83  vec3*2-vec3 (6), dot(xy,xy) (3)
    f += c["sphere_tests"] * 23        # :111-115 oc 3, a 5, halfB 5, c 7, discriminant 3
    f += c["disc_nonneg"] * 3          # :122-123 sqrt, sub, div
    f += c["far_roots"] * 2            # :127 add, div
    f += c["accepts"] * 18             # :136-145 p 6, n 6, dot 5, misc.x+0.5 1
    f += c["sky"] * 27                 # :263-266 normalize 9, t 2, mix 10, radiance 6
    f += c["sphere_iters"] * 11        # :63-65 vec3*2-1 (6), dot (5)
    f += c["lambert"] * 36             # :157-171 normalize 9, target 6, origin 6, normalize(target-p) 12, atten 3
    f += c["lambert_checker"] * 10     # :165-168 floor,floor,add 3, mod 4, albedo*tone 3
    f += c["metal"] * 41               # :201-207 normalize 9, reflect 12, origin 6, dir 6, atten 3, dot 5
    f += c["dielectric"] * 27          # :212-219,:230 atten 3, normalize 9, dot 5, sin 3, eta*sin 1, origin 6
    f += c["diel_front"] * 1           # :214 1.0 / refIdx
    f += c["diel_schlick"] * 9         # :194-196 r0 3, r0*r0 1, 1-cos 1, pow 1, 1-r0 1, mul 1, add 1
    f += (c["dielectric"] - c["diel_refract"]) * 12   # :223 reflect: dot 5, *2 1, *N 3, sub 3
    f += c["diel_refract"] * 27        # :178-188 dot 5, cos*n+i 6, eta* 3, perpDot 5, 1-perp 1, sqrt 1, *n 3, add 3
    f += c["pixels"] * 15              # :334-340 prev+accum 3, /count 3, c/(c+1) 6, pow 3
    return int(f)

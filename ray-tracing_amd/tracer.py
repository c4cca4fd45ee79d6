"""ctypes binding of librt_mi355x.so and a Python mirror of the reference's RayTracer class.

The class keeps the reference's method names and argument meaning
(src/rt/RayTracer.h:40-51) in snake_case: create, resize, destroy, set_camera,
set_samples_per_pixel, set_aperture, set_focus_distance, set_fov, set_max_depth,
render.  Errors come back as RuntimeError carrying the library's message, the way
VK_CHECK raises std::runtime_error in the reference (src/util/Check.h:8-19).

There is no CPU fallback: if the HIP library is missing, loading fails.
"""
import ctypes
import os

import numpy as np

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "librt_mi355x.so")


class GPUSphere(ctypes.Structure):
    """src/rt/RayTracer.h:12-17"""
    _fields_ = [("center_radius", ctypes.c_float * 4),
                ("albedo", ctypes.c_float * 4),
                ("misc", ctypes.c_float * 4)]


class GPUParams(ctypes.Structure):
    """src/rt/RayTracer.h:20-32"""
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


assert ctypes.sizeof(GPUSphere) == 48
assert ctypes.sizeof(GPUParams) == 160

# Every symbol include/rt_mi355x.h declares: name -> (restype, argtypes)
_H = ctypes.c_void_p
_U32 = ctypes.c_uint32
_F = ctypes.c_float
_F3 = ctypes.POINTER(ctypes.c_float)
_STREAM = ctypes.c_void_p
ABI = {
    "rt_create": (ctypes.c_int, [ctypes.c_int, _U32, _U32, ctypes.POINTER(_H)]),
    "rt_resize": (ctypes.c_int, [_H, _U32, _U32]),
    "rt_destroy": (ctypes.c_int, [_H]),
    "rt_set_camera": (ctypes.c_int, [_H, _F3, _F3, _F]),
    "rt_set_samples_per_pixel": (ctypes.c_int, [_H, _U32]),
    "rt_set_aperture": (ctypes.c_int, [_H, _F]),
    "rt_set_focus_distance": (ctypes.c_int, [_H, _F]),
    "rt_set_fov": (ctypes.c_int, [_H, _F]),
    "rt_set_max_depth": (ctypes.c_int, [_H, _U32]),
    "rt_set_spheres": (ctypes.c_int, [_H, ctypes.POINTER(GPUSphere), _U32]),
    "rt_get_spheres": (ctypes.c_int, [_H, ctypes.POINTER(GPUSphere), _U32, ctypes.POINTER(_U32)]),
    "rt_set_scene_hierarchy": (ctypes.c_int, [_H, ctypes.c_int]),
    "rt_get_scene_hierarchy": (ctypes.c_int, [_H, ctypes.c_void_p, ctypes.c_void_p, _U32, ctypes.POINTER(_U32)]),
    "rt_get_scene_hierarchy_links": (ctypes.c_int, [_H, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _U32,
                                                    ctypes.POINTER(_U32)]),
    "rt_get_scene_quadtree": (ctypes.c_int, [_H, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _U32, _U32, _U32,
                                             ctypes.POINTER(_U32), ctypes.POINTER(_U32), ctypes.POINTER(_U32),
                                             ctypes.POINTER(_U32), ctypes.POINTER(_U32), ctypes.POINTER(_U32),
                                             ctypes.c_void_p]),
    "rt_set_frame_overlap": (ctypes.c_int, [_H, ctypes.c_int]),
    "rt_get_params": (ctypes.c_int, [_H, _U32, ctypes.POINTER(GPUParams)]),
    "rt_reset_accumulation": (ctypes.c_int, [_H]),
    "rt_render": (ctypes.c_int, [_H, _U32, _STREAM]),
    "rt_render_frames": (ctypes.c_int, [_H, _U32, _U32, _STREAM]),
    "rt_render_shard": (ctypes.c_int, [_H, _U32, _U32, _U32, _U32, _U32, ctypes.c_int, _STREAM]),
    "rt_render_stripes": (ctypes.c_int, [_H, _U32, _U32, _U32, _U32, _U32, _U32, ctypes.c_int, _STREAM]),
    "rt_render_frames_static": (ctypes.c_int, [_H, _U32, _U32, _STREAM]),
    "rt_static_scratch_bytes": (ctypes.c_size_t, []),
    "rt_resolve_output": (ctypes.c_int, [_H, _U32, _U32, _U32, _STREAM]),
    "rt_resolve_output_from": (ctypes.c_int, [_H, ctypes.c_void_p, _U32, _U32, _U32, _STREAM]),
    "rt_accum_device_ptr": (ctypes.c_void_p, [_H]),
    "rt_output_device_ptr": (ctypes.c_void_p, [_H]),
    "rt_read_accum": (ctypes.c_int, [_H, ctypes.c_void_p, ctypes.c_size_t, _STREAM]),
    "rt_write_accum": (ctypes.c_int, [_H, ctypes.c_void_p, ctypes.c_size_t, _STREAM]),
    "rt_read_output": (ctypes.c_int, [_H, ctypes.c_void_p, ctypes.c_size_t, _STREAM]),
    "rt_enable_rng_capture": (ctypes.c_int, [_H, ctypes.c_int]),
    "rt_read_rng_state": (ctypes.c_int, [_H, ctypes.c_void_p, ctypes.c_size_t, _STREAM]),
    "rt_debug_fail_allocation": (None, [ctypes.c_int]),
    "rt_synchronize": (ctypes.c_int, [_H, _STREAM]),
    "rt_get_extent": (ctypes.c_int, [_H, ctypes.POINTER(_U32), ctypes.POINTER(_U32)]),
    "rt_last_kernel_ms": (ctypes.c_float, [_H]),
    "rt_kernel_ms": (ctypes.c_float, [_H, _U32]),
    "rt_last_error": (ctypes.c_char_p, [_H]),
    "rt_probe_hit_sphere": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(GPUSphere), ctypes.c_void_p, _U32,
                                           ctypes.c_void_p, ctypes.c_void_p]),
    "rt_probe_pow": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, _U32, ctypes.c_void_p]),
    "rt_probe_arith": (ctypes.c_int, [ctypes.c_int, _U32, ctypes.c_void_p, ctypes.c_void_p, _U32, ctypes.c_void_p,
                                      ctypes.c_void_p]),
    "rt_probe_rng": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, _U32, _U32, ctypes.c_void_p]),
}

_lib = None


def _preload_pytorch_hip_runtime():
    """PyTorch ships its own copy of the HIP runtime (torch/lib/libamdhip64.so).  A process can only use one
    copy: whichever is loaded first finds the GPU, the other reports none.  Where PyTorch is installed its copy
    is therefore loaded before this library (without importing torch), so that the tracer and a later
    `import torch` share it whatever the import order.  RT_MI355X_SYSTEM_HIP=1 skips this."""
    if os.environ.get("RT_MI355X_SYSTEM_HIP"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        candidate = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(candidate):
            ctypes.CDLL(candidate, mode=ctypes.RTLD_GLOBAL)
    except (ImportError, OSError, ValueError):
        pass


def load_library(path=None):
    """Loads librt_mi355x.so and declares every entry point.  Raises if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    _preload_pytorch_hip_runtime()
    # RT_MI355X_LIB selects another build of the same library (A/B timing of kernel variants)
    p = path or os.environ.get("RT_MI355X_LIB") or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            "HIP library %s is missing: build it with `make -C ray-tracing_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback" % p)
    lib = ctypes.CDLL(p)
    for name, (restype, argtypes) in ABI.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    if path is None:
        _lib = lib
    return lib


def _f3(v):
    return (ctypes.c_float * 3)(float(v[0]), float(v[1]), float(v[2]))


def spheres_to_ctypes(spheres):
    arr = np.ascontiguousarray(spheres, dtype=np.float32).reshape(-1, 12)
    buf = (GPUSphere * arr.shape[0]).from_buffer_copy(arr.tobytes())
    return buf, arr.shape[0]


class RayTracer:
    """Python mirror of class RayTracer (src/rt/RayTracer.h:34-94) over the C ABI."""

    def __init__(self, lib=None):
        """lib: another build of the library from load_library(path) (A/B comparisons); default: the shipped one."""
        self._lib = lib if lib is not None else load_library()
        self._h = _H()
        self._stream = None

    # -- plumbing ---------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            msg = self._lib.rt_last_error(self._h if self._h else None)
            raise RuntimeError("rt_mi355x error %d: %s" % (rc, (msg or b"").decode()))

    def set_stream(self, stream):
        """HIP stream handle (an int, e.g. torch.cuda.current_stream().cuda_stream) for render calls."""
        self._stream = ctypes.c_void_p(stream) if stream else None

    @property
    def handle(self):
        return self._h

    # -- RayTracer.h:40-51 --------------------------------------------------
    def create(self, width, height, device_id=0):
        """RayTracer::create: the extent replaces the swapchain the reference reads it from."""
        if self._h:
            raise RuntimeError("tracer already created")
        h = _H()
        rc = self._lib.rt_create(int(device_id), int(width), int(height), ctypes.byref(h))
        if rc != 0:
            msg = self._lib.rt_last_error(None)
            raise RuntimeError("rt_mi355x error %d: %s" % (rc, (msg or b"").decode()))
        self._h = h
        return self

    def resize(self, width, height):
        self._check(self._lib.rt_resize(self._h, int(width), int(height)))

    def destroy(self):
        if self._h:
            self._lib.rt_destroy(self._h)
            self._h = _H()

    def set_camera(self, pos, direction, focus_dist=-1.0):
        self._check(self._lib.rt_set_camera(self._h, _f3(pos), _f3(direction), float(focus_dist)))

    def set_samples_per_pixel(self, spp):
        self._check(self._lib.rt_set_samples_per_pixel(self._h, int(spp)))

    def set_aperture(self, aperture):
        self._check(self._lib.rt_set_aperture(self._h, float(aperture)))

    def set_focus_distance(self, focus_dist):
        self._check(self._lib.rt_set_focus_distance(self._h, float(focus_dist)))

    def set_fov(self, vfov):
        self._check(self._lib.rt_set_fov(self._h, float(vfov)))

    def set_max_depth(self, depth):
        self._check(self._lib.rt_set_max_depth(self._h, int(depth)))

    def render(self, frame_index):
        """RayTracer::render for one frame index (asynchronous on the stream)."""
        self._check(self._lib.rt_render(self._h, int(frame_index) & 0xFFFFFFFF, self._stream))

    # -- beyond the reference's public surface ----------------------------
    def set_spheres(self, spheres):
        buf, n = spheres_to_ctypes(spheres)
        self._check(self._lib.rt_set_spheres(self._h, buf, n))

    def get_spheres(self):
        n = _U32()
        self._check(self._lib.rt_get_spheres(self._h, None, 0, ctypes.byref(n)))
        buf = (GPUSphere * n.value)()
        self._check(self._lib.rt_get_spheres(self._h, buf, n.value, ctypes.byref(n)))
        return np.frombuffer(bytes(buf), dtype=np.float32).reshape(-1, 12).copy()

    def set_scene_hierarchy(self, enable):
        """Scenes of 24 spheres or more are traced through a bounding-sphere hierarchy (same images);
        enable=False tests every sphere against every ray like the shader's loop (raytrace.comp.glsl:247-259);
        enable=2 walks the depth-first tree, one node per step, also where the four-way tree would be used."""
        self._check(self._lib.rt_set_scene_hierarchy(self._h, int(enable) if enable in (0, 1, 2) else (1 if enable else 0)))

    def get_scene_quadtree(self):
        """The current scene's complete four-way tree of bounding boxes: dict(box (first_leaf, 6, 4) float32 -- lo.x,
        lo.y, lo.z, hi.x, hi.y, hi.z of the four children of a group above the last level --, geom (groups - first_leaf,
        4, 4) float32 -- the groups of four spheres (centre, radius^2) of the last level, then the prologue --,
        slot_sphere (slots,) uint32, groups, levels, roots, first_leaf, prologue_spheres, bound_centre, bound_radius,
        min_radius); groups == 0 when the scene has no such tree."""
        n_groups, n_slots, levels, roots, first_leaf, prologue = _U32(), _U32(), _U32(), _U32(), _U32(), _U32()
        refs = [ctypes.byref(v) for v in (n_groups, n_slots, levels, roots, first_leaf, prologue)]
        bound = np.zeros(5, dtype=np.float32)
        self._check(self._lib.rt_get_scene_quadtree(self._h, None, None, None, 0, 0, 0, *refs, bound.ctypes.data))
        n_box = first_leaf.value if n_groups.value else 0
        n_geom = n_groups.value - n_box
        box = np.zeros((n_box, 6, 4), dtype=np.float32)
        geom = np.zeros((n_geom, 4, 4), dtype=np.float32)
        slot_sphere = np.zeros((n_slots.value,), dtype=np.uint32)
        if n_groups.value:
            self._check(self._lib.rt_get_scene_quadtree(self._h, box.ctypes.data, geom.ctypes.data, slot_sphere.ctypes.data,
                                                        n_box, n_geom, n_slots.value, *refs, bound.ctypes.data))
        return {"box": box, "geom": geom, "slot_sphere": slot_sphere, "groups": n_groups.value, "levels": levels.value,
                "roots": roots.value, "first_leaf": first_leaf.value, "prologue_spheres": prologue.value,
                "bound_centre": bound[:3].copy(), "bound_radius": float(bound[3]), "min_radius": float(bound[4])}

    def set_frame_overlap(self, enable):
        """Consecutive single-frame calls overlap on the GPU (default); False keeps every launch on the
        caller's stream."""
        self._check(self._lib.rt_set_frame_overlap(self._h, 1 if enable else 0))

    def get_scene_hierarchy(self):
        """(geom (n, 4) float32, word (n,) uint32) of the current scene's hierarchy; n = 0 if it has none."""
        n = _U32()
        self._check(self._lib.rt_get_scene_hierarchy(self._h, None, None, 0, ctypes.byref(n)))
        geom = np.zeros((n.value, 4), dtype=np.float32)
        word = np.zeros((n.value,), dtype=np.uint32)
        if n.value:
            self._check(self._lib.rt_get_scene_hierarchy(self._h, geom.ctypes.data, word.ctypes.data, n.value,
                                                         ctypes.byref(n)))
        return geom, word

    def get_scene_hierarchy_links(self):
        """(geom (n, 4) float32, link (n,) uint32, sphere (n,) uint32) of the current scene's hierarchy, for
        trees of any size: link = node to continue with when a node's test fails, sphere = sphere index of
        a leaf or 0xFFFFFFFF."""
        n = _U32()
        self._check(self._lib.rt_get_scene_hierarchy_links(self._h, None, None, None, 0, ctypes.byref(n)))
        geom = np.zeros((n.value, 4), dtype=np.float32)
        link = np.zeros((n.value,), dtype=np.uint32)
        sphere = np.zeros((n.value,), dtype=np.uint32)
        if n.value:
            self._check(self._lib.rt_get_scene_hierarchy_links(self._h, geom.ctypes.data, link.ctypes.data,
                                                               sphere.ctypes.data, n.value, ctypes.byref(n)))
        return geom, link, sphere

    def get_params(self, frame_index=0):
        p = GPUParams()
        self._check(self._lib.rt_get_params(self._h, int(frame_index), ctypes.byref(p)))
        return p

    def get_params_words(self, frame_index=0):
        return np.frombuffer(bytes(self.get_params(frame_index)), dtype=np.uint32).copy()

    def reset_accumulation(self):
        self._check(self._lib.rt_reset_accumulation(self._h))

    def render_frames(self, first_frame, n_frames):
        self._check(self._lib.rt_render_frames(self._h, int(first_frame), int(n_frames), self._stream))

    def render_shard(self, first_frame, n_frames, frame_stride=1, row_begin=0, row_end=None, write_output=True):
        if row_end is None:
            row_end = self.extent[1]
        self._check(self._lib.rt_render_shard(self._h, int(first_frame), int(n_frames), int(frame_stride),
                                              int(row_begin), int(row_end), 1 if write_output else 0, self._stream))

    def render_stripes(self, first_frame, n_frames, frame_stride=1, stripe_rows=8, stripe_first=0, stripe_stride=1,
                       write_output=True):
        """Every stripe_stride-th stripe of stripe_rows rows, starting at stripe stripe_first (rank r of G
        passes stripe_first=r, stripe_stride=G)."""
        self._check(self._lib.rt_render_stripes(self._h, int(first_frame), int(n_frames), int(frame_stride),
                                                int(stripe_rows), int(stripe_first), int(stripe_stride),
                                                1 if write_output else 0, self._stream))

    def render_frames_static(self, first_frame, n_frames):
        """render_frames without any device allocation: the work buffers live in the library's static
        device array (see rt_render_frames_static)."""
        self._check(self._lib.rt_render_frames_static(self._h, int(first_frame), int(n_frames), self._stream))

    def resolve_output(self, sample_count, row_begin=0, row_end=None):
        if row_end is None:
            row_end = self.extent[1]
        self._check(self._lib.rt_resolve_output(self._h, int(sample_count), int(row_begin), int(row_end), self._stream))

    def resolve_output_from(self, accum_device_ptr, sample_count, row_begin=0, row_end=None):
        if row_end is None:
            row_end = self.extent[1]
        self._check(self._lib.rt_resolve_output_from(self._h, ctypes.c_void_p(accum_device_ptr), int(sample_count),
                                                     int(row_begin), int(row_end), self._stream))

    @property
    def extent(self):
        w, h = _U32(), _U32()
        self._check(self._lib.rt_get_extent(self._h, ctypes.byref(w), ctypes.byref(h)))
        return w.value, h.value

    def accum_device_ptr(self):
        return self._lib.rt_accum_device_ptr(self._h)

    def output_device_ptr(self):
        return self._lib.rt_output_device_ptr(self._h)

    def read_accum(self):
        w, h = self.extent
        out = np.empty((h, w, 4), dtype=np.float32)
        self._check(self._lib.rt_read_accum(self._h, out.ctypes.data, out.nbytes, self._stream))
        return out

    def write_accum(self, accum):
        w, h = self.extent
        a = np.ascontiguousarray(accum, dtype=np.float32)
        assert a.shape == (h, w, 4)
        self._check(self._lib.rt_write_accum(self._h, a.ctypes.data, a.nbytes, self._stream))

    def read_output(self):
        w, h = self.extent
        out = np.empty((h, w, 4), dtype=np.uint8)
        self._check(self._lib.rt_read_output(self._h, out.ctypes.data, out.nbytes, self._stream))
        return out

    def enable_rng_capture(self, enable=True):
        self._check(self._lib.rt_enable_rng_capture(self._h, 1 if enable else 0))

    def read_rng_state(self):
        w, h = self.extent
        out = np.empty((h, w), dtype=np.uint32)
        self._check(self._lib.rt_read_rng_state(self._h, out.ctypes.data, out.nbytes, self._stream))
        return out

    def synchronize(self):
        self._check(self._lib.rt_synchronize(self._h, self._stream))

    def last_kernel_ms(self):
        return float(self._lib.rt_last_kernel_ms(self._h))

    def kernel_ms(self, launches_ago=0):
        """GPU time of one of the last 64 trace launches (HIP events on the launch stream)."""
        return float(self._lib.rt_kernel_ms(self._h, int(launches_ago)))

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


# -- single-function device probes (vector tests) ---------------------------
def probe_hit_sphere(spheres, rays, device_id=0):
    """spheres (n,12) f32; rays (n,8) f32 = o.xyz,tmin,d.xyz,tmax -> (out (n,8) f32, front (n,) i32)."""
    lib = load_library()
    buf, n = spheres_to_ctypes(spheres)
    r = np.ascontiguousarray(rays, dtype=np.float32).reshape(n, 8)
    out = np.empty((n, 8), dtype=np.float32)
    front = np.empty((n,), dtype=np.int32)
    rc = lib.rt_probe_hit_sphere(device_id, buf, r.ctypes.data, n, out.ctypes.data, front.ctypes.data)
    if rc != 0:
        raise RuntimeError("rt_probe_hit_sphere failed: %s" % lib.rt_last_error(None).decode())
    return out, front


def probe_pow(x, y, device_id=0):
    lib = load_library()
    xa = np.ascontiguousarray(x, dtype=np.float32).ravel()
    ya = np.ascontiguousarray(y, dtype=np.float32).ravel()
    assert xa.size == ya.size
    out = np.empty_like(xa)
    rc = lib.rt_probe_pow(device_id, xa.ctypes.data, ya.ctypes.data, xa.size, out.ctypes.data)
    if rc != 0:
        raise RuntimeError("rt_probe_pow failed: %s" % lib.rt_last_error(None).decode())
    return out


def probe_arith(op, a, b=None, device_id=0):
    """op 0: a / b, 1: sqrt(a), 2: 1 / sqrt(a) -> (cheap-range helper result, full IEEE sequence result)."""
    lib = load_library()
    xa = np.ascontiguousarray(a, dtype=np.float32).ravel()
    xb = np.ascontiguousarray(b if b is not None else np.ones_like(xa), dtype=np.float32).ravel()
    assert xa.size == xb.size
    cheap = np.empty_like(xa)
    full = np.empty_like(xa)
    rc = lib.rt_probe_arith(device_id, op, xa.ctypes.data, xb.ctypes.data, xa.size, cheap.ctypes.data, full.ctypes.data)
    if rc != 0:
        raise RuntimeError("rt_probe_arith failed: %s" % lib.rt_last_error(None).decode())
    return cheap, full


def probe_rng(xyz, draws, device_id=0):
    lib = load_library()
    a = np.ascontiguousarray(xyz, dtype=np.uint32).reshape(-1, 3)
    out = np.empty((a.shape[0], 1 + draws), dtype=np.uint32)
    rc = lib.rt_probe_rng(device_id, a.ctypes.data, a.shape[0], draws, out.ctypes.data)
    if rc != 0:
        raise RuntimeError("rt_probe_rng failed: %s" % lib.rt_last_error(None).decode())
    return out

"""ctypes binding of librt_mi355x_rccl.so (include/rt_mi355x_rccl.h): the multi-GPU leg behind the C ABI.

A C++ host calls these entry points directly; this module exists so that bench.py and the tests drive
the same native code path.  The communicator is the library's own (ncclCommInitRank on an id that rank 0
creates and the host program distributes -- here over torch.distributed, whatever its backend).
"""
import ctypes
import os

from . import tracer

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "librt_mi355x_rccl.so")
ID_BYTES = 128

_H = ctypes.c_void_p
_C = ctypes.c_void_p
_U32 = ctypes.c_uint32
_STREAM = ctypes.c_void_p
ABI = {
    "rt_comm_unique_id": (ctypes.c_int, [ctypes.c_char_p]),
    "rt_comm_create": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_C)]),
    "rt_comm_adopt": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_C)]),
    "rt_comm_destroy": (ctypes.c_int, [_C]),
    "rt_comm_world": (ctypes.c_int, [_C]),
    "rt_comm_rank": (ctypes.c_int, [_C]),
    "rt_gather_stripes": (ctypes.c_int, [_H, _C, _U32, ctypes.c_void_p, _STREAM]),
    "rt_reduce_frames": (ctypes.c_int, [_H, _C, ctypes.c_void_p, _STREAM]),
    "rt_comm_pack_stripes": (ctypes.c_int, [_H, _C, _U32, _U32, _STREAM]),
    "rt_comm_gather_packed": (ctypes.c_int, [_H, _C, _U32, _U32, ctypes.c_void_p, _STREAM]),
    "rt_comm_stage_accum": (ctypes.c_int, [_H, _C, _U32, _STREAM]),
    "rt_comm_reduce_staged": (ctypes.c_int, [_H, _C, _U32, ctypes.POINTER(ctypes.c_void_p), _STREAM]),
    "rt_stripes_packed_bytes": (ctypes.c_size_t, [_U32, _U32, _U32, _U32]),
    "rt_pack_stripes_to": (ctypes.c_int, [_H, _U32, _U32, _U32, ctypes.c_void_p, _STREAM]),
    "rt_unpack_stripes_from": (ctypes.c_int, [_H, _U32, _U32, ctypes.c_void_p, ctypes.c_void_p, _STREAM]),
    "rt_comm_last_error": (ctypes.c_char_p, [_C]),
}

_lib = None


def _preload_pytorch_rccl():
    """PyTorch ships its own copy of RCCL (torch/lib/librccl.so, the same soname as ROCm's).  A process should
    run one copy: where PyTorch is installed its copy is loaded first, so that this library, which asks the
    loader for librccl.so.1, binds to it whatever the import order."""
    if os.environ.get("RT_MI355X_SYSTEM_HIP"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        candidate = os.path.join(list(spec.submodule_search_locations)[0], "lib", "librccl.so")
        if os.path.exists(candidate):
            ctypes.CDLL(candidate, mode=ctypes.RTLD_GLOBAL)
    except (ImportError, OSError, ValueError):
        pass


def load_library():
    """Loads librt_mi355x_rccl.so (after the tracer's library, which it calls) and declares every entry point."""
    global _lib
    if _lib is None:
        tracer.load_library()
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing: build it with `make -C ray-tracing_amd/csrc`" % LIB_PATH)
        _preload_pytorch_rccl()
        lib = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in ABI.items():
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
    return _lib


def unique_id():
    """128 bytes that rank 0 creates and every rank passes to Communicator."""
    lib = load_library()
    buf = ctypes.create_string_buffer(ID_BYTES)
    rc = lib.rt_comm_unique_id(buf)
    if rc != 0:
        raise RuntimeError("rt_comm_unique_id failed (%d): %s" % (rc, (lib.rt_comm_last_error(None) or b"").decode()))
    return buf.raw


def unique_id_over_torch(rank, device=None):
    """The id of rank 0, handed to every rank through torch.distributed (any backend)."""
    import torch
    import torch.distributed as dist
    raw = unique_id() if rank == 0 else bytes(ID_BYTES)
    t = torch.tensor(list(raw), dtype=torch.uint8)
    if dist.get_backend() == "nccl":
        t = t.to(device)
    dist.broadcast(t, src=0)
    return bytes(t.cpu().tolist())


def _check_global(lib, rc):
    if rc != 0:
        raise RuntimeError("rt_mi355x_rccl error %d: %s" % (rc, (lib.rt_comm_last_error(None) or b"").decode()))


def stripes_packed_bytes(width, height, stripe_rows, world):
    return int(load_library().rt_stripes_packed_bytes(int(width), int(height), int(stripe_rows), int(world)))


def pack_stripes_to(tracer_obj, stripe_rows, world, rank, send_ptr, stream=None):
    """The pack kernel alone: stripes rank, rank + world, ... of the tracer's accumulator into device memory at send_ptr."""
    lib = load_library()
    _check_global(lib, lib.rt_pack_stripes_to(tracer_obj.handle, int(stripe_rows), int(world), int(rank), ctypes.c_void_p(send_ptr),
                                              ctypes.c_void_p(stream) if stream else None))


def unpack_stripes_from(tracer_obj, stripe_rows, world, recv_ptr, dst_ptr, stream=None):
    """The unpack kernel alone: the ranks' packed buffers (rank-major at recv_ptr) into image order at dst_ptr."""
    lib = load_library()
    _check_global(lib, lib.rt_unpack_stripes_from(tracer_obj.handle, int(stripe_rows), int(world), ctypes.c_void_p(recv_ptr),
                                                  ctypes.c_void_p(dst_ptr), ctypes.c_void_p(stream) if stream else None))


class Communicator:
    """One rank's end of an RCCL communicator plus the library's staging buffers."""

    def __init__(self, id_bytes, world, rank, device_id=0):
        self._lib = load_library()
        self._c = _C()
        rc = self._lib.rt_comm_create(id_bytes, int(world), int(rank), int(device_id), ctypes.byref(self._c))
        if rc != 0:
            raise RuntimeError("rt_comm_create failed (%d): %s" % (rc, (self._lib.rt_comm_last_error(None) or b"").decode()))
        self.world, self.rank = int(world), int(rank)

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError("rt_mi355x_rccl error %d: %s" % (rc, (self._lib.rt_comm_last_error(self._c) or b"").decode()))

    @staticmethod
    def _stream(stream):
        return ctypes.c_void_p(stream) if stream else None

    def gather_stripes(self, tracer_obj, stripe_rows, dst_ptr, stream=None):
        self._check(self._lib.rt_gather_stripes(tracer_obj.handle, self._c, int(stripe_rows), ctypes.c_void_p(dst_ptr), self._stream(stream)))

    def reduce_frames(self, tracer_obj, dst_ptr, stream=None):
        self._check(self._lib.rt_reduce_frames(tracer_obj.handle, self._c, ctypes.c_void_p(dst_ptr), self._stream(stream)))

    def pack_stripes(self, tracer_obj, stripe_rows, slot, stream=None):
        self._check(self._lib.rt_comm_pack_stripes(tracer_obj.handle, self._c, int(stripe_rows), int(slot), self._stream(stream)))

    def gather_packed(self, tracer_obj, stripe_rows, slot, dst_ptr, stream=None):
        self._check(self._lib.rt_comm_gather_packed(tracer_obj.handle, self._c, int(stripe_rows), int(slot), ctypes.c_void_p(dst_ptr),
                                                    self._stream(stream)))

    def stage_accum(self, tracer_obj, slot, stream=None):
        self._check(self._lib.rt_comm_stage_accum(tracer_obj.handle, self._c, int(slot), self._stream(stream)))

    def reduce_staged(self, tracer_obj, slot, stream=None):
        """Returns the device address of the reduced image (the staging buffer)."""
        out = ctypes.c_void_p()
        self._check(self._lib.rt_comm_reduce_staged(tracer_obj.handle, self._c, int(slot), ctypes.byref(out), self._stream(stream)))
        return out.value

    def destroy(self):
        if self._c:
            self._lib.rt_comm_destroy(self._c)
            self._c = _C()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

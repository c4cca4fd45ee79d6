"""Sharding of one image over the GPUs of a node (one process per GPU, torch.distributed).

The reference renders on one device (src/vk/VulkanContext.cpp:413-414); pixels are independent
(raytrace.comp.glsl:333-335 touches only the invocation's own texel) and the random stream of a
texel depends only on (x, y, frame) (:310-311), so there are two exact ways to split the work:

* row bands:   rank r renders rows [b_r, b_{r+1}) of every frame; the bands are then gathered
               (all_gather of contiguous float4 slabs).  Bit-identical to one GPU.
* frame split: rank r renders frames r, r+G, r+2G, ... of the whole image into its own zeroed
               accumulator; the accumulators are summed (all_reduce, fp32) and alpha is set back
               to 1 (the shader stores alpha = 1 every frame, :335).  RNG streams are identical to
               one GPU; the sum order differs, i.e. a few ulp.

Everything here works on CPU tensors with the gloo backend too, which is how it is tested without GPUs.
"""
import torch
import torch.distributed as dist


def row_bands(height, world_size, align=8):
    """Contiguous row bands [(begin, end)] of near-equal size, multiples of `align` rows (the
    wave tile height) except possibly the last."""
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    tiles = (height + align - 1) // align
    bands = []
    begin = 0
    for r in range(world_size):
        n_tiles = tiles // world_size + (1 if r < tiles % world_size else 0)
        end = min(height, begin + n_tiles * align)
        bands.append((begin, end))
        begin = end
    return bands


def frame_shard(first_frame, n_frames_per_rank, world_size, rank):
    """(first, count, stride) of the frames rank `rank` renders when `world_size` ranks interleave
    frame indices starting at first_frame."""
    return first_frame + rank, n_frames_per_rank, world_size


def gather_row_bands(accum, bands, group=None):
    """accum: (H, W, 4) float32 tensor whose rows bands[rank] are valid.  After the call every rank
    holds the complete image.  Bands may differ in size, so each band is broadcast from its owner
    (over xGMI that is one direct transfer per peer)."""
    world = dist.get_world_size(group)
    assert len(bands) == world
    work = []
    for r, (b, e) in enumerate(bands):
        if e > b:
            work.append(dist.broadcast(accum[b:e], src=dist.get_global_rank(group, r) if group else r,
                                       group=group, async_op=True))
    for w in work:
        w.wait()
    return accum


def reduce_frame_split(accum, out=None, group=None):
    """Sums the per-rank accumulators of a frame split.  accum: (H, W, 4) float32.  Returns the
    reduced image (in `out` if given, else in place) with alpha restored to 1."""
    if out is None:
        out = accum
    else:
        out.copy_(accum)
    dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
    out[..., 3] = 1.0
    return out


class DeviceImage:
    """Exposes a device allocation of the tracer as a torch tensor without copying
    (through __cuda_array_interface__)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def accum_tensor(tracer, device):
    w, h = tracer.extent
    return torch.as_tensor(DeviceImage(tracer.accum_device_ptr(), (h, w, 4), "<f4"), device=device)


def output_tensor(tracer, device):
    w, h = tracer.extent
    return torch.as_tensor(DeviceImage(tracer.output_device_ptr(), (h, w, 4), "|u1"), device=device)

"""Sharding of one image over the GPUs of a node (one process per GPU, torch.distributed).

The reference renders on one device (src/vk/VulkanContext.cpp:413-414); pixels are independent
(raytrace.comp.glsl:333-335 touches only the invocation's own texel) and the random stream of a
texel depends only on (x, y, frame) (:310-311), so there are two exact ways to split the work:

* row bands:   rank r renders rows [b_r, b_{r+1}) of every frame; the bands are then gathered
               (all_gather of contiguous float4 slabs).  Bit-identical to one GPU.
* stripes:     the same with interleaved rows: the image is cut into stripes of a few rows and rank r
               renders stripes r, r+G, r+2G, ... (rt_render_stripes), so that cheap sky rows and
               expensive ground rows are spread evenly; one all_gather_into_tensor of equal-sized
               packed shards and a de-interleaving copy put the image together.  Bit-identical too.
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


def stripe_count(height, stripe_rows):
    return (height + stripe_rows - 1) // stripe_rows


def stripes_of_rank(height, stripe_rows, world_size, rank):
    """Indices of the stripes rank `rank` renders: rank, rank + world_size, ..."""
    return list(range(rank, stripe_count(height, stripe_rows), world_size))


def _all_gather_into(recv, send, group):
    """all_gather_into_tensor; device tensors on a backend without device support for it (gloo, used to
    rehearse several ranks on one GPU) go through host copies."""
    if send.is_cuda and dist.get_backend(group) == "gloo":
        host = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_gather_into_tensor(host, send.cpu(), group=group)
        recv.copy_(host)
    else:
        dist.all_gather_into_tensor(recv, send, group=group)


def _world_rank(group, rehearse):
    if rehearse is not None:
        return rehearse
    return dist.get_world_size(group), dist.get_rank(group)


def pack_stripes(accum, stripe_rows, group=None, send=None, rehearse=None):
    """Copies this rank's stripes (rank, rank + G, ... of stripes of stripe_rows rows; the last one may be cut
    short by the image) out of the (H, W, C) image into a (K, stripe_rows, W, C) buffer, K = ceil(stripes / G),
    on the current stream, and returns the buffer (`send` if given).  After it the image may be written again."""
    world, rank = _world_rank(group, rehearse)
    height, width, channels = accum.shape
    n_stripes = stripe_count(height, stripe_rows)
    slots = (n_stripes + world - 1) // world
    if send is None:
        send = accum.new_zeros((slots, stripe_rows, width, channels))
    assert tuple(send.shape) == (slots, stripe_rows, width, channels)
    full_rounds = (height // stripe_rows) // world      # rounds in which every rank owns a complete stripe
    rows_in_rounds = full_rounds * world * stripe_rows
    if full_rounds:
        send[:full_rounds].copy_(accum[:rows_in_rounds].view(full_rounds, world, stripe_rows, width, channels)[:, rank])
    for stripe in range(full_rounds * world, n_stripes):
        if stripe % world == rank:
            rows = accum[stripe * stripe_rows:min(height, (stripe + 1) * stripe_rows)]
            send[stripe // world, :rows.shape[0]].copy_(rows)
    return send


def gather_packed(send, out, stripe_rows, group=None, rehearse=None):
    """The collective half: one all_gather_into_tensor of the ranks' packed stripes and the copy that puts
    them back in image order into `out` (H, W, C) -- for the stripes of complete rounds a single strided copy.
    Runs on the current stream and only reads `send`."""
    world, rank = _world_rank(group, rehearse)
    height, width, channels = out.shape
    n_stripes = stripe_count(height, stripe_rows)
    slots = (n_stripes + world - 1) // world
    full_rounds = (height // stripe_rows) // world
    rows_in_rounds = full_rounds * world * stripe_rows
    # (the collective wants the ranks' contributions concatenated along the first dimension)
    flat = send.new_empty((world * slots, stripe_rows, width, channels))
    if rehearse is not None:
        for r in range(world):
            flat[r * slots:(r + 1) * slots].copy_(send)
    else:
        _all_gather_into(flat, send, group)
    recv = flat.view(world, slots, stripe_rows, width, channels)
    if full_rounds:
        out[:rows_in_rounds].view(full_rounds, world, stripe_rows, width, channels).copy_(
            recv[:, :full_rounds].permute(1, 0, 2, 3, 4))
    for stripe in range(full_rounds * world, n_stripes):
        rows = out[stripe * stripe_rows:min(height, (stripe + 1) * stripe_rows)]
        rows.copy_(recv[stripe % world, stripe // world, :rows.shape[0]])
    return out


def gather_stripes(accum, stripe_rows, group=None, out=None, packed=None, rehearse=None):
    """accum: (H, W, C) tensor whose stripes rank, rank + G, ... are valid on this rank.  After the call `out`
    (default: accum itself) holds the complete image on every rank: pack_stripes, then gather_packed, on the
    current stream; bit-identical to the image of one GPU.

    With a separate `out`, accum is only read while the shard is packed: `packed`, a torch.cuda.Event, is
    recorded right after that.  (A caller that pipelines steps should rather pack on the stream that renders
    and run gather_packed on a side stream, so that the rendering stream never waits for the side stream:
    two streams that wait for each other can stall for good when the runtime makes them share a hardware
    queue.  bench.py does that.)

    rehearse = (world, rank): one process on one GPU stands in for that rank of a larger job, without a
    process group; the collective is replaced by device copies of this rank's buffer into every rank's
    place (the same bytes land in the same buffers, the image is not a valid one).  For timing rehearsals."""
    world, rank = _world_rank(group, rehearse)
    target = accum if out is None else out
    if world == 1:
        if target is not accum:
            target.copy_(accum)
        if packed is not None:
            packed.record()
        return target
    send = pack_stripes(accum, stripe_rows, group=group, rehearse=rehearse)
    if packed is not None:
        packed.record()
    return gather_packed(send, target, stripe_rows, group=group, rehearse=rehearse)


def reduce_in_place(image, group=None, rehearse=None):
    """The collective half of the frame split: sums the ranks' (H, W, 4) float32 images in place and restores
    alpha to 1 (the shader stores alpha = 1 every frame, raytrace.comp.glsl:335; a sum gives G)."""
    if rehearse is not None:
        # one process standing in for a rank of a larger job (timing rehearsal): device traffic of the
        # order of a reduce-scatter + all-gather of the image instead of the collective
        image.add_(image.clone())
    elif image.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal of several ranks on one GPU: gloo's own device path hangs on a side stream here
        host = image.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        image.copy_(host)
    else:
        dist.all_reduce(image, op=dist.ReduceOp.SUM, group=group)
    image[..., 3] = 1.0
    return image


def reduce_frame_split(accum, out=None, group=None, packed=None, rehearse=None):
    """Sums the per-rank accumulators of a frame split.  accum: (H, W, 4) float32.  Returns the
    reduced image (in `out` if given, else in place) with alpha restored to 1.  With a separate `out`,
    accum is only read by the copy that starts the call; `packed` (a torch.cuda.Event) is recorded after it."""
    if out is None:
        out = accum
    else:
        out.copy_(accum)
    if packed is not None:
        packed.record()
    return reduce_in_place(out, group=group, rehearse=rehearse)


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

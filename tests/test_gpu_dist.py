"""The multi-GPU path on real hardware, as far as one GPU allows: the RCCL leg on a one-rank process group
(tensors that alias the tracer's device images, the sum-reduction, the resolve of a reduced image), and the
sharded renders plus their collectives with two rank processes that share the GPU (gloo carries the
collective there; RCCL refuses two ranks on one device).  Needs a real MI355X."""
import os
import socket
import sys

import numpy as np
import pytest

from test_gpu_parity import make_tracer, oracle_frames, same

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP, DEPTH, FRAMES = 160, 100, 2, 8, 6


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_rccl_leg_on_a_one_rank_group(pkg, oracle):
    import torch
    import torch.distributed as dist
    from raytracing_amd import dist as rtdist
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group(backend="nccl", device_id=device, rank=0, world_size=1)
    try:
        t = make_tracer(pkg, W, H, SPP, DEPTH)
        t.set_stream(torch.cuda.current_stream(device).cuda_stream)
        accum = rtdist.accum_tensor(t, device)
        output = rtdist.output_tensor(t, device)
        assert accum.shape == (H, W, 4) and accum.dtype == torch.float32 and accum.data_ptr() == t.accum_device_ptr()
        # the aliasing tensor sees what the kernel wrote
        t.render_shard(0, FRAMES, 1, 0, H, write_output=False)
        torch.cuda.synchronize(device)
        got = t.read_accum()
        assert same(accum.cpu().numpy(), got)
        ref = oracle_frames(oracle, t, pkg.default_scene(), range(FRAMES))
        assert same(got, ref["accum"])
        # sum-reduction over the (one-rank) RCCL group into a second buffer, alpha back to 1
        accum[..., 3] = 5.0
        reduced = torch.empty_like(accum)
        rtdist.reduce_frame_split(accum, out=reduced)
        torch.cuda.synchronize(device)
        assert same(reduced.cpu().numpy(), ref["accum"])
        # the display tail on the reduced buffer == the oracle's image of the same frames
        t.resolve_output_from(reduced.data_ptr(), FRAMES * SPP)
        torch.cuda.synchronize(device)
        assert same(t.read_output(), ref["output"])
        assert same(output.cpu().numpy(), ref["output"])
        # stripes on a one-rank group: the gather is the identity
        t.reset_accumulation()
        t.render_stripes(0, FRAMES, 1, 4, 0, 1, write_output=False)
        rtdist.gather_stripes(accum, 4)
        torch.cuda.synchronize(device)
        assert same(accum.cpu().numpy(), ref["accum"])
        # an all_gather_into_tensor on device memory the library owns
        flat = torch.empty((H, W, 4), dtype=torch.float32, device=device)
        dist.all_gather_into_tensor(flat, accum)
        torch.cuda.synchronize(device)
        assert same(flat.cpu().numpy(), ref["accum"])
        t.destroy()
    finally:
        dist.destroy_process_group()


def test_native_rccl_entry_points_on_a_one_rank_communicator(pkg, oracle):
    """The collectives behind the C ABI (librt_mi355x_rccl.so: hand-written pack / unpack kernels around
    ncclAllGather and ncclAllReduce) on a one-rank communicator of the library's own: the same checks as
    test_rccl_leg_on_a_one_rank_group makes of the torch.distributed path, bit for bit."""
    import torch
    from raytracing_amd import rccl
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    comm = rccl.Communicator(rccl.unique_id(), 1, 0, device_id=0)
    stream = torch.cuda.current_stream(device).cuda_stream
    t = make_tracer(pkg, W, H, SPP, DEPTH)
    t.set_stream(stream)
    ref = oracle_frames(oracle, t, pkg.default_scene(), range(FRAMES))
    # frame split: the (one-rank) sum into a second buffer, alpha back to 1
    t.render_shard(0, FRAMES, 1, 0, H, write_output=False)
    bad_alpha = t.read_accum()
    bad_alpha[..., 3] = 5.0
    t.write_accum(bad_alpha)
    reduced = torch.empty((H, W, 4), dtype=torch.float32, device=device)
    comm.reduce_frames(t, reduced.data_ptr(), stream)
    torch.cuda.synchronize(device)
    assert same(reduced.cpu().numpy(), ref["accum"])
    t.resolve_output_from(reduced.data_ptr(), FRAMES * SPP)
    assert same(t.read_output(), ref["output"])
    # ... in place, and through the two halves a pipelined caller uses
    comm.reduce_frames(t, t.accum_device_ptr(), stream)
    assert same(t.read_accum(), ref["accum"])
    t.write_accum(bad_alpha)
    comm.stage_accum(t, 1, stream)
    staged = comm.reduce_staged(t, 1, stream)
    t.resolve_output_from(staged, FRAMES * SPP)
    assert same(t.read_output(), ref["output"])
    # stripes: pack, all-gather, unpack -- heights that are no multiple of the stripe, in place and into a second image
    for stripe_rows in (4, 7, 64, 1000):
        t.reset_accumulation()
        t.render_stripes(0, FRAMES, 1, stripe_rows, 0, 1, write_output=False)
        image = torch.full((H, W, 4), 3.0, dtype=torch.float32, device=device)
        comm.gather_stripes(t, stripe_rows, image.data_ptr(), stream)
        torch.cuda.synchronize(device)
        assert same(image.cpu().numpy(), ref["accum"])
        comm.gather_stripes(t, stripe_rows, t.accum_device_ptr(), stream)
        assert same(t.read_accum(), ref["accum"])
    comm.pack_stripes(t, 4, 1, stream)
    image.fill_(0.0)
    comm.gather_packed(t, 4, 1, image.data_ptr(), stream)
    torch.cuda.synchronize(device)
    assert same(image.cpu().numpy(), ref["accum"])
    comm.destroy()
    # the second half without the first is an error, not a read of an unallocated buffer
    fresh = rccl.Communicator(rccl.unique_id(), 1, 0, device_id=0)
    with pytest.raises(RuntimeError, match="nothing was packed"):
        fresh.gather_packed(t, 4, 0, image.data_ptr(), stream)
    with pytest.raises(RuntimeError, match="no accumulator was staged"):
        fresh.reduce_staged(t, 0, stream)
    fresh.destroy()
    t.destroy()


@pytest.mark.parametrize("w,h,stripe_rows,world", [(160, 100, 4, 2), (96, 70, 8, 3), (64, 61, 4, 8), (50, 20, 16, 5), (33, 17, 1, 8),
                                                   (1920, 1080, 4, 8)])
def test_native_pack_and_unpack_kernels_for_several_ranks(pkg, w, h, stripe_rows, world):
    """The pack and unpack kernels of librt_mi355x_rccl.so with the layout of `world` ranks, on one GPU: every
    rank's stripes are rendered and packed into that rank's place of the buffer an all-gather would fill, and the
    unpacked image is the unsharded one bit for bit -- ragged last stripes and ranks without a stripe in the last
    round included."""
    import torch
    from raytracing_amd import rccl
    device = torch.device("cuda", 0)
    whole = make_tracer(pkg, w, h, 2, 6)
    whole.render_frames(0, 2)
    want = whole.read_accum()
    whole.destroy()
    per_rank = rccl.stripes_packed_bytes(w, h, stripe_rows, world)
    stripes = (h + stripe_rows - 1) // stripe_rows
    assert per_rank == ((stripes + world - 1) // world) * min(stripe_rows, h) * w * 16
    gathered = torch.full((world * per_rank // 4,), float("nan"), dtype=torch.float32, device=device)
    t = make_tracer(pkg, w, h, 2, 6)
    for rank in range(world):
        t.reset_accumulation()
        t.render_stripes(0, 2, 1, stripe_rows, rank, world, write_output=False)
        rccl.pack_stripes_to(t, stripe_rows, world, rank, gathered.data_ptr() + rank * per_rank)
    image = torch.full((h, w, 4), float("nan"), dtype=torch.float32, device=device)
    rccl.unpack_stripes_from(t, stripe_rows, world, gathered.data_ptr(), image.data_ptr())
    torch.cuda.synchronize(device)
    assert same(image.cpu().numpy(), want)
    assert not torch.isnan(gathered).any()          # the padding of the packed buffers is written too (zeros)
    t.destroy()


def _worker(rank, world, port, mode, result_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import rtamd_loader
    pkg = rtamd_loader.load()
    from raytracing_amd import dist as rtdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        device = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        t = pkg.RayTracer().create(W, H, device_id=0)
        t.set_samples_per_pixel(SPP)
        t.set_max_depth(DEPTH)
        t.set_stream(torch.cuda.current_stream(device).cuda_stream)
        accum = rtdist.accum_tensor(t, device)
        if mode == "tiles":
            t.render_stripes(0, FRAMES, 1, 8, rank, world, write_output=False)
            rtdist.gather_stripes(accum, 8)
            t.resolve_output(FRAMES * SPP)
            image = accum
        elif mode == "bands":
            bands = rtdist.row_bands(H, world)
            t.render_shard(0, FRAMES, 1, bands[rank][0], bands[rank][1], write_output=False)
            rtdist.gather_row_bands(accum, bands)
            t.resolve_output(FRAMES * SPP)
            image = accum
        else:
            t.render_shard(rank, len(range(rank, FRAMES, world)), world, 0, H, write_output=False)
            image = torch.empty_like(accum)
            rtdist.reduce_frame_split(accum, out=image)
            t.resolve_output_from(image.data_ptr(), FRAMES * SPP)
        torch.cuda.synchronize(device)
        if rank == 0:
            np.savez(result_path, accum=image.cpu().numpy(), output=t.read_output())
        t.destroy()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,world", [("tiles", 2), ("tiles", 3), ("bands", 2), ("frames", 2)])
def test_rank_processes_sharing_the_gpu(pkg, oracle, tmp_path, mode, world):
    import torch.multiprocessing as mp
    result = str(tmp_path / ("%s%d.npz" % (mode, world)))
    mp.spawn(_worker, args=(world, _free_port(), mode, result), nprocs=world, join=True)
    got = np.load(result)
    host = pkg.RayTracer().create(W, H, device_id=-1)
    host.set_samples_per_pixel(SPP)
    host.set_max_depth(DEPTH)
    ref = oracle_frames(oracle, host, pkg.default_scene(), range(FRAMES))
    assert np.all(got["accum"][..., 3] == 1.0)
    if mode == "frames":
        lum = float(ref["accum"][..., :3].mean())
        rmse = float(np.sqrt(np.mean((got["accum"][..., :3] - ref["accum"][..., :3]) ** 2)))
        assert rmse <= 1e-3 * lum
        assert np.abs(got["output"].astype(np.int32) - ref["output"].astype(np.int32)).max() <= 1
    else:
        assert same(got["accum"], ref["accum"])
        assert same(got["output"], ref["output"])


def test_bench_starts_its_own_ranks(tmp_path):
    """`bench.py --gpus 2` without a launcher: two rank processes, one JSON line.  (Both ranks share the
    one GPU of this box and gloo carries the collective; on a multi-GPU node the same command line minus
    the two rehearsal flags runs one rank per GPU over RCCL.)"""
    import json
    import subprocess
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--same-device", "--backend", "gloo", "--verify"], env=env, capture_output=True, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong" and rec["value"] > 0
    assert rec["config"]["samples_per_step"] == 1920 * 1080 * 16 * 64
    # the image the two ranks put together over four pipelined steps is the one a single GPU renders
    assert rec["verified_against_one_gpu"] == "bit-identical"


def test_bench_frame_split_of_three_ranks_verifies(tmp_path):
    """The frame split (sum all-reduce, alpha back to 1) through bench.py's pipelined step loop, three rank
    processes on this box's one GPU over gloo: the reduced image agrees with one GPU's to rounding."""
    import json
    import subprocess
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--mode", "frames", "--steps", "3",
                          "--warmup", "1", "--same-device", "--backend", "gloo", "--verify"], env=env, capture_output=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    rec = json.loads([l for l in out.stdout.decode().splitlines() if l.startswith("{")][0])
    assert rec["n_gpus"] == 3
    assert rec["verified_against_one_gpu"]["alpha_is_one"]
    assert rec["verified_against_one_gpu"]["rmse_over_mean"] < 1e-5

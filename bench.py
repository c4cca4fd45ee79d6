#!/usr/bin/env python3
"""Throughput benchmark of the traced path: path samples/sec at 1080p, depth 8.

One "step" renders the batch the headline metric is quoted on: the default five-sphere scene at
1920x1080, 16 samples per pixel per frame, depth 8, depth of field off, 64 accumulated frames
(2.12e9 path samples) -- one launch of the trace kernel per rank.  With --gpus N > 1 the frame
indices are interleaved over the ranks (each rank still renders 64 frames of the full image, i.e.
per-GPU work is fixed: weak scaling), followed by the step's real exchange: an RCCL sum all-reduce of
the float4 accumulators and the display resolve of the reduced image.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WIDTH, HEIGHT, SPP, DEPTH, FRAMES = 1920, 1080, 16, 8, 64
PEAK_FP32_VALU_TFLOPS = 157.3  # MI355X vector FP32: 256 CU x 256 FLOP/clk x 2.4 GHz


def load_flops_per_sample():
    """Algorithmic FLOP per path sample of this workload, from the instrumented oracle's event counters
    (profiles/flops_per_sample.json, produced by tools/count_flops.py)."""
    path = os.path.join(ROOT, "profiles", "flops_per_sample.json")
    with open(path) as f:
        return json.load(f)["c2_1080p_spp16_depth8_dof_off"]["flops_per_sample"]


def load_measured_traffic():
    """HBM bytes per trace launch from the committed rocprofv3 PMC passes of this same command
    (FETCH_SIZE doubled as gfx950 requires for wide reads, plus WRITE_SIZE); None if not recorded.
    Profiles of this command live in profiles/r<round><letter>/; directories with a suffix
    (r1i_c5m ...) hold other workloads."""
    import re
    best = None
    base = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(base)):
        path = os.path.join(base, name, "pmc_summary.json")
        if re.fullmatch(r"r\d+[a-z]*", name) and os.path.isfile(path):
            d = json.load(open(path)).get("_derived", {})
            if "hbm_read_bytes_per_launch" in d:
                best = d["hbm_read_bytes_per_launch"] + d["hbm_write_bytes_per_launch"]
    return best


def cpu_baseline(pkg, frames=24):
    """The CPU oracle (a port of the reference shader) timed on this machine's host cores on a bounded
    sample of the same workload."""
    from oracle import pyoracle
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16") or 16), 64)
    host = pkg.RayTracer().create(WIDTH, HEIGHT, device_id=-1)
    host.set_samples_per_pixel(SPP)
    host.set_max_depth(DEPTH)
    host.set_aperture(0.0)
    scene = pkg.default_scene()
    accum = None
    t0 = time.perf_counter()
    for f in range(frames):
        accum = pyoracle.render(host.get_params_words(f), scene, accum=accum, want_output=False,
                                nthreads=threads)["accum"]
    dt = time.perf_counter() - t0
    samples = WIDTH * HEIGHT * SPP * frames
    return {"value": samples / dt, "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": "frames 0..%d of the same 1080p/16spp/depth-8 workload (%.3g samples, %.1f s)" % (
                frames - 1, samples, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-distributed", action="store_true",
                    help="run the multi-GPU code path (process group, all-reduce, resolve) even with one rank")
    args = ap.parse_args()

    import torch
    import rtamd_loader
    pkg = rtamd_loader.load()
    from raytracing_amd import dist as rtdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("--gpus %d needs a torch.distributed launch with that many ranks" % args.gpus)
    distributed = world > 1 or args.force_distributed
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", device_id=device, rank=rank, world_size=world)

    tracer = pkg.RayTracer().create(WIDTH, HEIGHT, device_id=local_rank)
    tracer.set_samples_per_pixel(SPP)
    tracer.set_max_depth(DEPTH)
    tracer.set_aperture(0.0)  # depth of field off; the lens draws are still consumed
    stream = torch.cuda.current_stream(device)
    tracer.set_stream(stream.cuda_stream)

    accum = rtdist.accum_tensor(tracer, device) if distributed else None
    reduced = torch.empty_like(accum) if distributed else None

    def step(index):
        # rank r renders frames base + r, base + r + world, ... (64 of them)
        base = index * FRAMES * world
        first, count, stride = rtdist.frame_shard(base, FRAMES, world, rank)
        tracer.render_shard(first, count, stride, 0, HEIGHT, write_output=not distributed)
        if distributed:
            rtdist.reduce_frame_split(accum, out=reduced)
            total_frames = (index + 1) * FRAMES * world
            tracer.resolve_output_from(reduced.data_ptr(), total_frames * SPP)

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(device)

    for i in range(args.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    # GPU time of each timed step's trace launch: the library brackets every launch with HIP events
    # on the launch stream; reading them after the fence does not perturb the timed region
    kernel_ms = [tracer.kernel_ms(i) for i in range(min(args.steps, 64))]

    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    samples_per_rank_step = WIDTH * HEIGHT * SPP * FRAMES
    total_samples = samples_per_rank_step * world * args.steps
    value = total_samples / elapsed

    if rank == 0:
        flops_per_sample = load_flops_per_sample()
        avg_kernel_s = (sum(kernel_ms) / len(kernel_ms)) * 1e-3
        achieved = flops_per_sample * samples_per_rank_step / avg_kernel_s / 1e12
        result = {
            "metric": "path samples/sec at 1080p depth 8",
            "value": value,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "default 5-sphere scene, 1920x1080, 16 spp/frame, depth 8, DoF off, "
                                   "64 accumulated frames per GPU per step",
                       "parallelism": "frame-index split x%d + RCCL sum all-reduce" % world if distributed else "1 GPU",
                       "samples_per_step": samples_per_rank_step * world},
            "roofline": {"bound": "valu", "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_VALU_TFLOPS, "traffic": load_measured_traffic(),
                         "kernel_ms": avg_kernel_s * 1e3, "flops_per_sample": flops_per_sample,
                         "note": "FP32 VALU-bound kernel (no MFMA; HBM traffic is the 16 B per-frame sum per pixel "
                                 "per frame, ~2.1 GB per launch, i.e. ~50 GB/s); achieved = algorithmic FLOP per launch / "
                                 "HIP-event kernel time; traffic = bytes per launch from the committed rocprofv3 PMC passes"},
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(pkg)
        print(json.dumps(result))

    tracer.destroy()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

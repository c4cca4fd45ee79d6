#!/usr/bin/env python3
"""Throughput benchmark of the traced path: path samples/sec at 1080p, depth 8.

One "step" renders the batch the headline metric is quoted on (workload c2): the default five-sphere
scene at 1920x1080, 16 samples per pixel per frame, depth 8, depth of field off, 64 accumulated frames
(2.12e9 path samples).  On one GPU that is one launch of the trace kernel per step.

With --gpus N > 1 one process per GPU shares each step (strong scaling by default: the step's work is
fixed, N ranks split it) in one of the two ways the path shards:

  --mode tiles   (default)  interleaved row stripes of the image, every rank renders all frames of its
                            stripes, then ONE all_gather_into_tensor of the float4 accumulator shards
                            (RCCL) and the display resolve.  Bit-identical to one GPU.
  --mode frames             interleaved frame indices of the whole image, then a sum all-reduce of the
                            float4 accumulators, alpha back to 1, and the display resolve.

The collective and the resolve are inside the timed region.  --scaling weak (frames mode only) keeps 64
frames per rank instead.  --workload picks the other configurations: c3 (4K, 64 spp, depth 16, DoF on,
64 frames), c4 (8K, 256 spp, one frame; meant for --mode tiles), c5 (4096 spheres, depth 50, 32 frames;
meant for --mode frames).

Launch: `python bench.py --gpus N` starts its N rank processes itself; under a torch.distributed
launcher (WORLD_SIZE set) it is one of the ranks.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_VALU_TFLOPS = 157.3  # MI355X vector FP32: 256 CU x 256 FLOP/clk x 2.4 GHz
PEAK_LDS_GBPS = 78643.2        # LDS: 256 CU x 128 B/clk x 2.4 GHz

WORKLOADS = {
    # the configuration the metric is quoted on
    "c2": dict(width=1920, height=1080, spp=16, depth=8, aperture=0.0, frames=64, scene="default",
               flops_key="c2_1080p_spp16_depth8_dof_off",
               text="default 5-sphere scene, 1920x1080, 16 spp/frame, depth 8, DoF off, 64 accumulated frames"),
    "c3": dict(width=3840, height=2160, spp=64, depth=16, aperture=0.05, frames=64, scene="default",
               flops_key="c3_4k_spp64_depth16_dof_on",
               text="default 5-sphere scene, 3840x2160, 64 spp/frame, depth 16, DoF on, 64 accumulated frames (4096 spp)"),
    "c4": dict(width=7680, height=4320, spp=256, depth=8, aperture=0.05, frames=1, scene="default",
               flops_key="c4_8k_spp256_depth8_dof_on",
               text="default 5-sphere scene, 7680x4320, 256 spp, depth 8, DoF on, one frame"),
    # 32 frames per step (the configuration fixes no frame count): with 8 ranks that is four frames per rank and launch,
    # which amortises the ramp and drain of a hierarchy launch (one frame per rank projects to 6.1x, four to 7.1x)
    "c5": dict(width=1920, height=1080, spp=16, depth=50, aperture=0.05, frames=32, scene="synthetic4096",
               flops_key="c5_4096_spheres_depth50",
               text="synthetic 4096-sphere mixed-material scene, 1920x1080, 16 spp/frame, depth 50, 32 frames"),
}


def load_flops_per_sample(key):
    """Algorithmic FLOP per path sample of a workload, from the instrumented oracle's event counters
    (profiles/flops_per_sample.json, produced by tools/count_flops.py); None if not recorded."""
    path = os.path.join(ROOT, "profiles", "flops_per_sample.json")
    with open(path) as f:
        rec = json.load(f).get(key)
    return rec["flops_per_sample"] if rec else None


def load_hierarchy_bytes(key):
    """LDS bytes per path sample of a workload traced through the hierarchy (group visits of the four-way tree of
    boxes counted by the diagnostic build: 96 bytes per box group, 64 per group of four spheres), or None."""
    path = os.path.join(ROOT, "profiles", "flops_per_sample.json")
    with open(path) as f:
        rec = json.load(f).get(key) or {}
    return (rec.get("hierarchy") or {}).get("lds_bytes_per_sample")


def load_measured_traffic():
    """HBM bytes per trace launch of the c2 workload from the committed rocprofv3 PMC passes of this same
    command (FETCH_SIZE doubled as gfx950 requires for wide reads, plus WRITE_SIZE), with the file they were
    read from and the other per-launch figures of those passes; (None, None, {}) if not recorded.
    Profiles of this command live in profiles/r<round><letter>/; directories with a suffix
    (r1i_c5m ...) hold other workloads.  The most recent one wins: the figure describes the kernel as it
    was when that profile was taken, which is why its source is named in the bench line."""
    import re
    best, source, derived = None, None, {}
    base = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(base)):
        path = os.path.join(base, name, "pmc_summary.json")
        if re.fullmatch(r"r\d+[a-z]*", name) and os.path.isfile(path):
            d = json.load(open(path)).get("_derived", {})
            if "hbm_read_bytes_per_launch" in d:
                best = d["hbm_read_bytes_per_launch"] + d["hbm_write_bytes_per_launch"]
                source, derived = os.path.relpath(path, ROOT), d
    return best, source, derived


def load_uniform_lanes_probe():
    """What the same kernel reached on views whose lanes all do the same thing (tools/uniform_lanes_probe.py): the
    most recent committed profiles/r*/uniform_lanes_probe.json, as {view: algorithmic TFLOP/s}, and the file's path."""
    import re
    found, source = {}, None
    base = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(base)):
        path = os.path.join(base, name, "uniform_lanes_probe.json")
        if re.fullmatch(r"r\d+[a-z]*", name) and os.path.isfile(path):
            found = {view: rec["tflops"] for view, rec in json.load(open(path)).items()}
            source = os.path.relpath(path, ROOT)
    return found, source


def load_workload_pmc(workload):
    """Per-launch figures of the committed rocprofv3 PMC passes of another workload (profiles/r<round><letter>_<workload>/),
    the most recent one, with the file they were read from; ({}, None) if there is none."""
    import re
    derived, source = {}, None
    base = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(base)):
        path = os.path.join(base, name, "pmc_summary.json")
        if re.fullmatch(r"r\d+[a-z]*_" + re.escape(workload), name) and os.path.isfile(path):
            derived, source = json.load(open(path)).get("_derived", {}), os.path.relpath(path, ROOT)
    return derived, source


def host_threads():
    """Threads for the CPU baseline: every core this process may run on (its affinity mask; the machine's
    core count can be many times that on a shared box), bounded by OMP_NUM_THREADS where the environment
    sets a share."""
    try:
        allowed = len(os.sched_getaffinity(0))
    except AttributeError:
        allowed = os.cpu_count() or 1
    share = os.environ.get("OMP_NUM_THREADS")
    if share and share.isdigit() and int(share) > 0:
        allowed = min(allowed, int(share))
    return max(1, min(allowed, 256)), (os.cpu_count() or 1)


def cpu_baseline(pkg, wl, budget_s=10.0):
    """The CPU oracle (a port of the reference shader) timed on this machine's host cores: as many frames
    of the benchmark workload as fit the time budget, and the configuration quoted for a CPU run of the reference
    (1280x720, 1 spp, depth 8) next to it."""
    from oracle import pyoracle
    threads, machine = host_threads()
    scene = pkg.default_scene()

    def run(width, height, spp, depth, aperture, min_frames, seconds):
        host = pkg.RayTracer().create(width, height, device_id=-1)
        host.set_samples_per_pixel(spp)
        host.set_max_depth(depth)
        host.set_aperture(aperture)
        accum = None
        frames = 0
        t0 = time.perf_counter()
        while frames < min_frames or time.perf_counter() - t0 < seconds:
            accum = pyoracle.render(host.get_params_words(frames), scene, accum=accum, want_output=False,
                                    nthreads=threads)["accum"]
            frames += 1
        dt = time.perf_counter() - t0
        host.destroy()
        return width * height * spp * frames / dt, frames, dt

    rate, frames, dt = run(wl["width"], wl["height"], wl["spp"], wl["depth"], wl["aperture"], 2, budget_s)
    c1_rate, c1_frames, c1_dt = run(1280, 720, 1, 8, 0.05, 8, 1.0)
    result = {"value": rate, "unit": "samples/s", "cores": threads, "host_cpus": machine, "kind": "port",
              "sample": "frames 0..%d of the same workload (%.3g samples, %.1f s, %d threads of %d host CPUs)" % (
                  frames - 1, wl["width"] * wl["height"] * wl["spp"] * frames, dt, threads, machine),
              "c1": {"value": c1_rate, "unit": "samples/s",
                     "sample": "1280x720, 1 spp, depth 8, aperture 0.05, frames 0..%d (%.2f s)" % (c1_frames - 1, c1_dt)}}
    result["reference_on_lavapipe"] = lavapipe_baseline(pkg, wl, scene, threads, budget_s)
    if result["reference_on_lavapipe"].get("available"):
        # the reference's own shader ran on this machine's cores: that is the baseline, the port rides along
        ref = result["reference_on_lavapipe"]
        result["port"] = {"value": rate, "unit": "samples/s", "cores": threads, "sample": result["sample"]}
        result.update(value=ref["value"], kind="reference", cores=threads,
                      sample="%d frames of the same workload through shaders/raytrace.comp.glsl on %s (%.1f s)" % (
                          ref["frames"], ref["device"], ref["seconds"]))
    return result


def lavapipe_baseline(pkg, wl, scene, threads, budget_s):
    """The reference shader itself on Mesa lavapipe (oracle/lavapipe.py, oracle/_ref/lavapipe_harness): runs where the
    machine has a Vulkan loader, lavapipe and a SPIR-V build of the shader -- RT_LAVAPIPE_SPV names one, or a GLSL
    compiler on PATH builds it from RT_LAVAPIPE_SHADER (default: the reference tree).  Anywhere else the entry says
    what is missing; the figure is never estimated."""
    try:
        from oracle import lavapipe
        lavapipe.probe(timeout=20)
        import tempfile
        with tempfile.TemporaryDirectory(prefix="lavapipe_") as tmp:
            spv = os.environ.get("RT_LAVAPIPE_SPV") or lavapipe.compile_shader(
                os.environ.get("RT_LAVAPIPE_SHADER", lavapipe.DEFAULT_SHADER), os.path.join(tmp, "raytrace.comp.spv"))
            host = pkg.RayTracer().create(wl["width"], wl["height"], device_id=-1)
            host.set_samples_per_pixel(wl["spp"])
            host.set_max_depth(wl["depth"])
            host.set_aperture(wl["aperture"])
            words = host.get_params_words(0)
            host.destroy()
            os.environ.setdefault("LP_NUM_THREADS", str(threads))
            out = lavapipe.baseline(spv, words, scene, budget_s=budget_s)
        out["available"] = True
        return out
    except Exception as e:  # noqa: BLE001 -- Unavailable says what is missing; anything else must not stop the GPU leg
        return {"available": False, "reason": "%s: %s" % (type(e).__name__, e)}


def other_call_patterns(pkg, device_index, stream):
    """Outside the timed region: the same image rendered the way the reference's caller drives it
    (src/core/App.cpp:403: one render(frame) per frame; consecutive calls overlap on the GPU), with the
    benchmark's settings and with the application's defaults (4 spp, depth 12, aperture 0.05)."""
    out = {}
    for name, spp, depth, aperture in (("c2_one_render_call_per_frame", 16, 8, 0.0),
                                       ("application_defaults_one_render_call_per_frame", 4, 12, 0.05)):
        t = pkg.RayTracer().create(1920, 1080, device_id=device_index)
        t.set_samples_per_pixel(spp)
        t.set_max_depth(depth)
        t.set_aperture(aperture)
        t.set_stream(stream.cuda_stream)
        frames = 64
        for rep in range(2):
            t.synchronize()
            t0 = time.perf_counter()
            for f in range(frames):
                t.render(f)
            t.synchronize()
            dt = time.perf_counter() - t0
        out[name] = {"ms_per_frame": dt / frames * 1e3, "samples_per_s": 1920 * 1080 * spp * frames / dt,
                     "settings": "1920x1080, %d spp, depth %d, aperture %g, 64 rt_render calls" % (spp, depth, aperture)}
        t.destroy()
    return out


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n):
    """Starts the N rank processes of this script (this process has not touched the GPU and never does)
    and relays rank 0's output."""
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return max(abs(c) for c in codes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2")
    ap.add_argument("--frames", type=int, default=0, help="frames per step instead of the workload's own count")
    ap.add_argument("--mode", choices=["tiles", "frames"], default="tiles", help="how N > 1 ranks share a step")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--stripe-rows", type=int, default=0, help="rows per interleaved stripe in tiles mode (0: 4, or 16 at 8K)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extra call patterns (for profiling runs)")
    ap.add_argument("--force-distributed", action="store_true",
                    help="run the multi-GPU code path (process group, collective, resolve) even with one rank")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl")
    ap.add_argument("--collectives", choices=["native", "torch"], default="native",
                    help="native: the library's own RCCL entry points (librt_mi355x_rccl.so: hand-written pack / unpack kernels "
                         "around ncclAllGather / ncclAllReduce on a communicator of its own), used whenever the backend is nccl; "
                         "torch: torch.distributed collectives and strided tensor copies (always used with gloo and in rehearsals)")
    ap.add_argument("--watchdog", type=int, default=300,
                    help="seconds after which a rank that is still running prints its Python stacks and exits (0 = off)")
    ap.add_argument("--same-device", action="store_true",
                    help="rehearsal on a one-GPU box: every rank uses cuda:0 (with --backend gloo)")
    ap.add_argument("--rehearse-world", type=int, default=0,
                    help="timing rehearsal on one GPU: this process does rank 0's part of a job of that many ranks, with "
                         "the collective replaced by device copies of the same size (not a measurement of several GPUs)")
    ap.add_argument("--comm-priority", type=int, default=0, help="priority of the stream the collective runs on")
    ap.add_argument("--verify", action="store_true",
                    help="after the timed steps rank 0 renders all their frames again on its own GPU and compares the image "
                         "the ranks put together with it (tiles: bit for bit; frames: relative RMSE); on by itself whenever "
                         "more than one rank runs (a few seconds, outside the timed region)")
    ap.add_argument("--no-verify", action="store_true", help="skip that comparison in a run of several ranks")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    # stdout carries the one JSON line and nothing else: whatever libraries print (RCCL's version banner,
    # for one) is sent to stderr by pointing descriptor 1 there for the duration of the run
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    if args.watchdog > 0:
        import faulthandler
        faulthandler.dump_traceback_later(args.watchdog, exit=True)   # a hung collective must not hang the whole job

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit("--gpus %d but the launcher started %d ranks" % (args.gpus, world))
    wl = WORKLOADS[args.workload]
    if args.frames > 0 and args.frames != wl["frames"]:
        wl = dict(wl, frames=args.frames, text=wl["text"] + " [--frames %d: that many per step instead]" % args.frames)
    rehearse = (args.rehearse_world, 0) if args.rehearse_world > 1 else None
    if rehearse and world != 1:
        sys.exit("--rehearse-world is for a single process")
    distributed = world > 1 or args.force_distributed or rehearse is not None
    split_world, split_rank = rehearse if rehearse else (world, rank)
    if args.scaling == "weak" and args.mode != "frames":
        sys.exit("--scaling weak needs --mode frames (a fixed image cannot grow with the ranks)")

    # torch before the tracer's library: torch brings its own copy of the HIP runtime, and the copy a process
    # loads first is the one that finds the GPU
    import torch
    import rtamd_loader
    pkg = rtamd_loader.load()

    # the CPU leg first: it needs no GPU, and the GPU then works until the end of the run
    baseline = None
    if rank == 0 and world == 1 and not args.force_distributed and not args.no_cpu_baseline:
        baseline = cpu_baseline(pkg, wl)

    from raytracing_amd import dist as rtdist
    device_index = 0 if args.same_device else local_rank
    torch.cuda.set_device(device_index)
    device = torch.device("cuda", device_index)
    if distributed and not rehearse:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    width, height, spp, frames = wl["width"], wl["height"], wl["spp"], wl["frames"]
    tracer = pkg.RayTracer().create(width, height, device_id=device_index)
    tracer.set_samples_per_pixel(spp)
    tracer.set_max_depth(wl["depth"])
    tracer.set_aperture(wl["aperture"])  # 0 = depth of field off; the lens draws are still consumed
    if wl["scene"] == "synthetic4096":
        tracer.set_spheres(pkg.synthetic_scene(4096))
        tracer.set_camera([3.0, 1.5, 6.0], [-0.4, -0.2, -1.0], 6.0)
    stream = torch.cuda.current_stream(device)
    tracer.set_stream(stream.cuda_stream)
    if not distributed:
        # one launch at a time on one GPU, so that the HIP-event time of a launch (the roofline figure) is that
        # launch's alone; the benchmark's own launch is too large to be overlapped anyway, the single-frame and
        # small-batch workloads (c4, c5) would be
        tracer.set_frame_overlap(False)

    stripe_rows = args.stripe_rows or (16 if height >= 4000 else 4)
    native = None
    native_error = None
    if distributed and not rehearse and args.backend == "nccl" and args.collectives == "native":
        # The library's own communicator.  If any rank cannot set it up (library missing, ncclCommInitRank refused),
        # every rank falls back to the torch.distributed collectives together, and the line says so.
        try:
            from raytracing_amd import rccl as rtrccl
            rtrccl.load_library()
            native = rtrccl.Communicator(rtrccl.unique_id_over_torch(rank, device), world, rank, device_id=device_index)
        except Exception as e:  # noqa: BLE001 -- whatever it was, the other path still measures the job
            native_error = "%s: %s" % (type(e).__name__, e)
        ok = torch.tensor([1 if native is not None else 0], dtype=torch.int32, device=device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if native is not None:
                native.destroy()
                native = None
            native_error = native_error or "another rank could not create the communicator"
            if rank == 0:
                print("bench.py: native collectives unavailable (%s); using torch.distributed" % native_error, file=sys.stderr)
    accum = rtdist.accum_tensor(tracer, device) if distributed else None
    # The collective of a step runs on a side stream, and the rendering stream never waits for that stream: the
    # rank's shard is copied out of the accumulator on the rendering stream itself (into one of two staging
    # buffers), the side stream waits for that copy, runs the collective, puts the image together in a second
    # image and resolves it for display, and before a staging buffer is used again the HOST waits for the
    # collective that read it two steps earlier.  So the next step's render starts right after the copy and runs
    # beside the collective, and no two streams ever wait for each other (streams that do can stall for good
    # when the runtime makes them share a hardware queue).
    image = torch.empty_like(accum) if distributed else None
    comm = torch.cuda.Stream(device, priority=args.comm_priority) if distributed else None
    staging = [None, None]
    collected = [None, None]
    frames_per_step = frames * split_world if args.scaling == "weak" else frames

    def step(index):
        base = index * frames_per_step
        if not distributed:
            tracer.render_frames(base, frames)
            return
        slot = index % 2
        if collected[slot] is not None:
            collected[slot].synchronize()      # host: the collective of two steps ago has read this staging buffer
        if native is not None:
            # the same pipeline through the library's own entry points: the shard is packed (or the accumulator
            # staged) on the rendering stream, the collective, the unpacking and the resolve run on the side stream
            if args.mode == "tiles":
                tracer.render_stripes(base, frames, 1, stripe_rows, rank, world, write_output=False)
                native.pack_stripes(tracer, stripe_rows, slot, stream.cuda_stream)
            else:
                count = len(range(rank, frames_per_step, world))
                if count:
                    tracer.render_shard(base + rank, count, world, 0, height, write_output=False)
                native.stage_accum(tracer, slot, stream.cuda_stream)
            copied = torch.cuda.Event()
            copied.record(stream)
            with torch.cuda.stream(comm):
                comm.wait_event(copied)
                tracer.set_stream(comm.cuda_stream)
                if args.mode == "tiles":
                    native.gather_packed(tracer, stripe_rows, slot, image.data_ptr(), comm.cuda_stream)
                    shown_ptr = image.data_ptr()
                else:
                    shown_ptr = native.reduce_staged(tracer, slot, comm.cuda_stream)
                    staging[slot] = shown_ptr
                tracer.resolve_output_from(shown_ptr, (base + frames_per_step) * spp)
                tracer.set_stream(stream.cuda_stream)
                collected[slot] = torch.cuda.Event()
                collected[slot].record(comm)
            return
        if args.mode == "tiles":
            # rank r: stripes r, r + world, ... of every frame of the step
            tracer.render_stripes(base, frames, 1, stripe_rows, split_rank, split_world, write_output=False)
            if split_world > 1:
                staging[slot] = rtdist.pack_stripes(accum, stripe_rows, send=staging[slot], rehearse=(split_world, split_rank))
            else:
                staging[slot] = accum.clone() if staging[slot] is None else staging[slot].copy_(accum)
        else:
            # rank r: frames base + r, base + r + world, ... of the whole image
            count = len(range(split_rank, frames_per_step, split_world))
            if count:
                tracer.render_shard(base + split_rank, count, split_world, 0, height, write_output=False)
            staging[slot] = accum.clone() if staging[slot] is None else staging[slot].copy_(accum)
        copied = torch.cuda.Event()
        copied.record(stream)
        with torch.cuda.stream(comm):
            comm.wait_event(copied)
            tracer.set_stream(comm.cuda_stream)
            if args.mode == "tiles":
                if split_world > 1:
                    rtdist.gather_packed(staging[slot], image, stripe_rows, rehearse=rehearse)   # the image is put together
                    shown = image
                else:
                    shown = staging[slot]
            else:
                shown = rtdist.reduce_in_place(staging[slot], rehearse=rehearse)                 # the partial sums are added
            tracer.resolve_output_from(shown.data_ptr(), (base + frames_per_step) * spp)
            tracer.set_stream(stream.cuda_stream)
            collected[slot] = torch.cuda.Event()
            collected[slot].record(comm)

    def fence():
        if distributed:
            comm.synchronize()
            if rehearse:
                pass
            elif args.backend == "nccl":
                dist.barrier(device_ids=[device_index])
            else:
                torch.cuda.synchronize(device)
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
    kernel_ms = [k for k in kernel_ms if k >= 0.0]

    if rehearse:
        kernel_spread = (sum(kernel_ms) / max(1, len(kernel_ms)),) * 2
    elif distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # the spread of the ranks' kernel times is the load imbalance of the split
        k = torch.tensor([sum(kernel_ms) / max(1, len(kernel_ms))], dtype=torch.float64, device=device)
        k_max, k_min = k.clone(), k.clone()
        dist.all_reduce(k_max, op=dist.ReduceOp.MAX)
        dist.all_reduce(k_min, op=dist.ReduceOp.MIN)
        kernel_spread = (float(k_min.item()), float(k_max.item()))

    samples_per_step = width * height * spp * frames_per_step
    value = samples_per_step * args.steps / elapsed

    verified = None
    if (args.verify or world > 1) and not args.no_verify and distributed and rank == 0 and not rehearse:
        import numpy as np
        last = (args.warmup + args.steps - 1) % 2
        if native is not None:
            together = image if args.mode == "tiles" else torch.as_tensor(
                rtdist.DeviceImage(staging[last], (height, width, 4), "<f4"), device=device)
            together = together.cpu().numpy()
        else:
            together = (staging[last] if (args.mode == "frames" or split_world == 1) else image).cpu().numpy()
        whole = pkg.RayTracer().create(width, height, device_id=device_index)
        whole.set_samples_per_pixel(spp)
        whole.set_max_depth(wl["depth"])
        whole.set_aperture(wl["aperture"])
        if wl["scene"] == "synthetic4096":
            whole.set_spheres(pkg.synthetic_scene(4096))
            whole.set_camera([3.0, 1.5, 6.0], [-0.4, -0.2, -1.0], 6.0)
        whole.render_frames(0, (args.warmup + args.steps) * frames_per_step)
        alone = whole.read_accum()
        whole.destroy()
        if args.mode == "tiles":
            verified = "bit-identical" if np.array_equal(alone.view(np.uint32), together.view(np.uint32)) else "DIFFERENT"
        else:
            rmse = float(np.sqrt(np.mean((alone[..., :3].astype(np.float64) - together[..., :3]) ** 2)))
            verified = {"rmse_over_mean": rmse / float(alone[..., :3].mean()), "alpha_is_one": bool(np.all(together[..., 3] == 1.0))}

    if rank == 0:
        flops_per_sample = load_flops_per_sample(wl["flops_key"])
        avg_kernel_s = (sum(kernel_ms) / len(kernel_ms)) * 1e-3
        # samples this rank's launch traced (its share of the step)
        if not distributed:
            launch_samples = samples_per_step
        elif args.mode == "tiles":
            my_rows = sum(min(height, (s + 1) * stripe_rows) - s * stripe_rows
                          for s in rtdist.stripes_of_rank(height, stripe_rows, split_world, split_rank))
            launch_samples = width * my_rows * spp * frames
        else:
            launch_samples = width * height * spp * len(range(split_rank, frames_per_step, split_world))
        if distributed:
            how = ("interleaved %d-row stripes x%d + RCCL all-gather of the accumulator" % (stripe_rows, split_world)
                   if args.mode == "tiles" else "frame-index split x%d + RCCL sum all-reduce of the accumulator" % split_world)
            how += (" (librt_mi355x_rccl.so: native pack/unpack kernels, ncclAllGather/ncclAllReduce)" if native is not None
                    else " (torch.distributed, backend %s%s)" % (args.backend, "; native collectives unavailable: " + native_error
                                                                  if native_error else ""))
            if rehearse:
                how = ("REHEARSAL ON ONE GPU, not a measurement of %d GPUs: rank 0's part of [%s], the collective replaced by "
                       "device copies of the same size; value = what %d ranks would reach if each took as long"
                       % (split_world, how, split_world))
        else:
            how = "1 GPU"
        result = {
            "metric": ("REHEARSAL of %d ranks on one GPU, not a measurement: " % split_world if rehearse else "") +
                      "path samples/sec at 1080p depth 8",
            "value": value,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s: %s" % (args.workload, wl["text"]) +
                                   (" per GPU" if args.scaling == "weak" and world > 1 else " per step"),
                       "parallelism": how,
                       "samples_per_step": samples_per_step},
        }
        lds_bytes = load_hierarchy_bytes(wl["flops_key"])
        if lds_bytes is not None:
            # The hierarchy kernel skips 99.9 % of the sphere tests the shader's loop makes, so the FLOP count "as
            # written" says nothing about it (it would come out above the peak).  What the walk moves is LDS: every
            # group visit is a read of 96 or 64 bytes at a per-lane address.
            achieved = lds_bytes * launch_samples / avg_kernel_s / 1e9
            pmc, pmc_source = load_workload_pmc(args.workload)
            result["roofline"] = {
                "bound": "lds", "achieved": achieved, "peak": PEAK_LDS_GBPS, "unit": "GB/s", "frac": achieved / PEAK_LDS_GBPS,
                "traffic": None, "kernel_ms": avg_kernel_s * 1e3, "lds_bytes_per_sample": lds_bytes,
                "flops_per_sample_as_written": flops_per_sample,
                "valu": {"insts_per_wave_sample": pmc.get("valu_insts_per_wave_sample"),
                         "lanes_per_inst": pmc.get("mean_active_lanes_per_valu_inst"),
                         "simd_cycles_per_valu_inst": pmc.get("simd_cycles_per_valu_inst"),
                         "waves_per_simd": pmc.get("mean_waves_per_simd"), "source": pmc_source},
                "note": "hierarchy walk (four-way tree of boxes in LDS): achieved = group visits per sample (diagnostic build) "
                        "x 96 B per box group / 64 B per group of spheres x samples per launch / HIP-event kernel time; peak = "
                        "256 CU x 128 B/clk x 2.4 GHz of LDS bandwidth.  The fraction is low because the walk is bound by VALU "
                        "issue at the 4 waves per SIMD its LDS footprint allows (one 1024-thread workgroup per CU holds the "
                        "tree; the PMC figures of the latest committed profile of this workload are in `valu`), "
                        "not by LDS bytes; HBM traffic is negligible"}
        elif flops_per_sample is not None:
            achieved = flops_per_sample * launch_samples / avg_kernel_s / 1e12
            traffic, traffic_source, pmc = load_measured_traffic() if args.workload == "c2" and not distributed else (None, None, {})
            result["roofline"] = {
                "bound": "valu", "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_VALU_TFLOPS,
                "traffic": traffic, "traffic_source": traffic_source,
                "lanes_per_inst": pmc.get("mean_active_lanes_per_valu_inst"),
                "valu_insts_per_wave_sample": pmc.get("valu_insts_per_wave_sample"),
                "simd_cycles_per_valu_inst": pmc.get("simd_cycles_per_valu_inst"),
                "kernel_ms": avg_kernel_s * 1e3, "flops_per_sample": flops_per_sample,
                "same_kernel_with_uniform_lanes": dict(zip(("tflops_by_view", "source"), load_uniform_lanes_probe())),
                "ceilings": {"unit": "TFLOP/s", "fma_every_slot": PEAK_FP32_VALU_TFLOPS,
                             "one_flop_per_slot": PEAK_FP32_VALU_TFLOPS / 2,
                             "one_flop_per_slot_at_measured_issue_rate": PEAK_FP32_VALU_TFLOPS / 2 * 2.0 / 2.45,
                             "note": "results must equal the shader's operation by operation, so no multiply-add is fused and a "
                                     "slot carries one flop; packed FP32 issues at twice the cycles of the plain instruction here "
                                     "(tools/ubench/valu_rates.hip) and adds nothing; 2.45 SIMD-cycles per VALU instruction is the "
                                     "PMC figure of this kernel with the VALU saturated (profiles/r2f/pmc_summary.json)"},
                "note": "FP32 VALU-bound kernel (no MFMA; HBM traffic is the 16 B per-frame sum per pixel per frame, "
                        "i.e. well under 100 GB/s); achieved = algorithmic FLOP per launch / HIP-event kernel time of "
                        "rank 0's launch; traffic, lanes_per_inst (active lanes of 64 per VALU instruction) and the two "
                        "figures after it are per-launch means from the committed rocprofv3 PMC passes named in "
                        "traffic_source, not from this run"}
        else:
            result["roofline"] = {"bound": "valu", "achieved": None, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                                  "frac": None, "traffic": None, "kernel_ms": avg_kernel_s * 1e3,
                                  "note": "no algorithmic FLOP count recorded for this workload (tools/count_flops.py)"}
        if distributed:
            result["rank_kernel_ms"] = {"min": kernel_spread[0], "max": kernel_spread[1],
                                        "note": "HIP-event time of each rank's trace launches; shard launches are small, so the "
                                                "library lets consecutive ones overlap and each time includes the shared part"}
        elif args.workload == "c2" and not args.no_extras:
            result["other_call_patterns"] = other_call_patterns(pkg, device_index, stream)
        if verified is not None:
            result["verified_against_one_gpu"] = verified
        if baseline is not None:
            result["cpu_baseline"] = baseline
        json_out.write(json.dumps(result) + "\n")
        json_out.flush()

    if native is not None:
        torch.cuda.synchronize(device)
        native.destroy()
    tracer.destroy()
    if distributed and not rehearse:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

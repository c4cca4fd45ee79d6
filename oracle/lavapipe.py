#!/usr/bin/env python3
"""The reference shader itself on the host's cores (Mesa lavapipe) -- TEST INFRASTRUCTURE ONLY.

Drives oracle/_ref/lavapipe_harness (oracle/lavapipe_harness.cpp): a headless Vulkan compute host that runs a
SPIR-V build of shaders/raytrace.comp.glsl with the bindings and the dispatch of src/rt/RayTracer.cpp:344-403,
:559-661.  Three things have to be on the machine for that: a Vulkan loader (libvulkan.so.1), a software device
behind it (lavapipe reports itself as "llvmpipe (LLVM ...)") and either a ready SPIR-V module or a GLSL compiler
(glslangValidator or glslc) plus the shader source.  Where one of them is missing every entry point here says which
and gives up; nothing is emulated.

  python oracle/lavapipe.py probe
  python oracle/lavapipe.py compare [--spv FILE | --shader FILE] [--width W --height H --spp S --depth D
                                    --aperture A --frames N] [--against oracle|hip] [--max-rmse X]

compare renders the default scene with the shader on lavapipe and with the CPU oracle (or the HIP library on
cuda:0) from the same 160-byte uniform block and sphere bytes, and prints: relative RMSE of the accumulators
(the acceptance bound for converged images is 1e-3 of the mean luminance at 4096 samples per pixel), the share of
bit-identical texels, and the differences of the two rgba8 images.  Exit status 77: not available here.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "oracle", "_ref", "lavapipe_harness")
DEFAULT_SHADER = "/root/reference/shaders/raytrace.comp.glsl"
EXIT_UNAVAILABLE = 77


class Unavailable(RuntimeError):
    """Something the run needs is not on this machine; str() says what."""


def _run_harness(args, timeout):
    if not os.path.isfile(HARNESS):
        raise Unavailable("oracle/_ref/lavapipe_harness is not built (make -C oracle harness, needs the reference's Vulkan headers)")
    done = subprocess.run([HARNESS] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    text = done.stdout.decode(errors="replace").strip()
    try:
        report = json.loads(text.splitlines()[-1]) if text else {}
    except ValueError:
        report = {}
    if done.returncode == EXIT_UNAVAILABLE:
        raise Unavailable(report.get("reason") or "the harness found no Vulkan device")
    if done.returncode != 0:
        raise RuntimeError("lavapipe_harness failed (%d): %s" % (done.returncode, done.stderr.decode(errors="replace").strip()[-400:]))
    return report


def probe(device="llvmpipe", timeout=30):
    """The harness's view of the machine: {"available": True, "device": ..., "devices": [...]}; raises Unavailable."""
    return _run_harness(["--probe", "--device", device], timeout)


def find_glsl_compiler():
    """(program, argument list up to the input file) of a GLSL-to-SPIR-V compiler on PATH, or None."""
    path = shutil.which("glslangValidator") or shutil.which("glslang")
    if path:
        return path, ["-V", "--target-env", "vulkan1.0", "-S", "comp", "-o"]
    path = shutil.which("glslc")
    if path:
        return path, ["-fshader-stage=compute", "--target-env=vulkan1.0", "-O0", "-o"]
    return None


def compile_shader(shader, out_spv):
    """Compiles the reference's GLSL source as it is (no optimisation passes of ours) into out_spv."""
    found = find_glsl_compiler()
    if found is None:
        raise Unavailable("no GLSL compiler (glslangValidator / glslc) on PATH, and no --spv given")
    if not os.path.isfile(shader):
        raise Unavailable("shader source %s not found" % shader)
    program, head = found
    subprocess.run([program] + head + [out_spv, shader], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return out_spv


def run(spv, params_words, spheres, first_frame=0, frames=1, device="llvmpipe", workdir=None, timeout=3600):
    """Runs `frames` consecutive dispatches from a cleared accumulator.  params_words: the 40 uint32 of the uniform
    block; spheres: (N, 12) float32.  Returns (report, accum (H, W, 4) float32, rgba8 (H, W, 4) uint8)."""
    import numpy as np
    words = np.ascontiguousarray(params_words, dtype=np.uint32)
    assert words.size == 40
    width, height = int(words[32:33].view(np.float32)[0]), int(words[33:34].view(np.float32)[0])
    own = workdir is None
    workdir = workdir or tempfile.mkdtemp(prefix="lavapipe_")
    try:
        p = lambda name: os.path.join(workdir, name)  # noqa: E731
        words.tofile(p("params.bin"))
        np.ascontiguousarray(spheres, dtype=np.float32).reshape(-1, 12).tofile(p("spheres.bin"))
        report = _run_harness(["--spv", spv, "--params", p("params.bin"), "--spheres", p("spheres.bin"),
                               "--first-frame", str(first_frame), "--frames", str(frames), "--device", device,
                               "--out-accum", p("accum.f32"), "--out-rgba8", p("shown.rgba8")], timeout)
        accum = np.fromfile(p("accum.f32"), dtype=np.float32).reshape(height, width, 4)
        shown = np.fromfile(p("shown.rgba8"), dtype=np.uint8).reshape(height, width, 4)
        return report, accum, shown
    finally:
        if own:
            shutil.rmtree(workdir, ignore_errors=True)


def baseline(spv, params_words, spheres, budget_s=10.0, device="llvmpipe"):
    """Throughput of the reference shader on lavapipe for bench.py's cpu_baseline leg: one untimed frame (the
    driver compiles the shader when it first runs), then as many frames as the harness's own timing of that
    frame says fit the budget.  Returns {"value": samples/s, "device", "frames", "seconds", "threads"}."""
    warm, _, _ = run(spv, params_words, spheres, frames=1, device=device)
    per_frame = max(warm["seconds"], 1e-6)
    frames = max(1, min(64, int(budget_s / per_frame)))
    report, _, _ = run(spv, params_words, spheres, frames=frames, device=device)
    return {"value": report["samples_per_s"], "unit": "samples/s", "device": report["device"], "frames": frames,
            "seconds": report["seconds"], "threads": report.get("LP_NUM_THREADS") or "driver default"}


def _cmd_probe(args):
    print(json.dumps(probe(args.device)))
    return 0


def _cmd_compare(args):
    import numpy as np
    sys.path.insert(0, ROOT)
    import rtamd_loader
    pkg = rtamd_loader.load()
    from oracle import pyoracle

    probe(args.device)
    workdir = tempfile.mkdtemp(prefix="lavapipe_")
    try:
        spv = args.spv or compile_shader(args.shader, os.path.join(workdir, "raytrace.comp.spv"))
        host = pkg.RayTracer().create(args.width, args.height, device_id=-1 if args.against == "oracle" else 0)
        host.set_samples_per_pixel(args.spp)
        host.set_max_depth(args.depth)
        host.set_aperture(args.aperture)
        scene = pkg.default_scene()
        report, accum, shown = run(spv, host.get_params_words(0), scene, frames=args.frames, device=args.device, workdir=workdir)
        if args.against == "oracle":
            ours = None
            for frame in range(args.frames):
                res = pyoracle.render(host.get_params_words(frame), scene, accum=ours, want_output=True)
                ours = res["accum"]
            ours_shown = res["output"]
        else:
            host.render_frames(0, args.frames)
            ours, ours_shown = host.read_accum(), host.read_output()
        host.destroy()
        count = args.frames * args.spp
        rmse, relative = pkg.image.rmse_linear(accum, count, ours, count)
        out = {"lavapipe": report, "against": args.against, "samples_per_pixel": count, "rmse": rmse, "relative_rmse": relative,
               "identical_texels": float(np.mean(np.all(accum.view(np.uint32) == ours.view(np.uint32), axis=-1))),
               "rgba8": pkg.image.diff_rgba8(shown, ours_shown)}
        print(json.dumps(out))
        return 1 if args.max_rmse is not None and relative > args.max_rmse else 0
    finally:
        shutil.rmtree(workdir, ignore_errors=True)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="command", required=True)
    p = sub.add_parser("probe")
    p.add_argument("--device", default="llvmpipe")
    p.set_defaults(fn=_cmd_probe)
    c = sub.add_parser("compare")
    c.add_argument("--spv")
    c.add_argument("--shader", default=DEFAULT_SHADER)
    c.add_argument("--device", default="llvmpipe")
    c.add_argument("--width", type=int, default=1280)
    c.add_argument("--height", type=int, default=720)
    c.add_argument("--spp", type=int, default=1)
    c.add_argument("--depth", type=int, default=8)
    c.add_argument("--aperture", type=float, default=0.05)
    c.add_argument("--frames", type=int, default=1)
    c.add_argument("--against", choices=["oracle", "hip"], default="oracle")
    c.add_argument("--max-rmse", type=float)
    c.set_defaults(fn=_cmd_compare)
    args = ap.parse_args(argv)
    try:
        return args.fn(args)
    except Unavailable as e:
        print(json.dumps({"available": False, "reason": str(e)}))
        return EXIT_UNAVAILABLE


if __name__ == "__main__":
    sys.exit(main())

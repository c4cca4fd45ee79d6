#!/usr/bin/env python3
"""Renders a scene through the C ABI on one GPU and writes the display image (PNG or PPM) and,
optionally, the linear accumulator (npz).

  python tools/render.py out.png [--scene default|random|synthetic4096|FILE] [--size 1280x720]
         [--spp 16] [--depth 8] [--aperture 0.05] [--fov 20] [--frames 16]
         [--camera=px,py,pz --look=dx,dy,dz [--focus f]] [--accum out.npz]   (use = so that negative numbers are not taken for options)
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("output")
    ap.add_argument("--scene", default="default")
    ap.add_argument("--size", default="1280x720")
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--aperture", type=float, default=0.05)
    ap.add_argument("--fov", type=float, default=20.0)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--camera")
    ap.add_argument("--look")
    ap.add_argument("--focus", type=float, default=-1.0)
    ap.add_argument("--accum")
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args()

    import rtamd_loader
    pkg = rtamd_loader.load()
    from raytracing_amd import image

    w, h = (int(v) for v in args.size.lower().split("x"))
    t = pkg.RayTracer().create(w, h, device_id=args.device)
    t.set_samples_per_pixel(args.spp)
    t.set_max_depth(args.depth)
    t.set_aperture(args.aperture)
    t.set_fov(args.fov)
    if args.scene == "random":
        t.set_spheres(pkg.random_scene())
    elif args.scene == "synthetic4096":
        t.set_spheres(pkg.synthetic_scene(4096))
    elif args.scene != "default":
        t.set_spheres(pkg.load_scene(args.scene))
    if args.camera or args.look:
        if not (args.camera and args.look):
            ap.error("--camera and --look go together")
        pos = [float(v) for v in args.camera.split(",")]
        look = [float(v) for v in args.look.split(",")]
        t.set_camera(pos, look, args.focus)
    elif args.focus > 0.0:
        t.set_focus_distance(args.focus)
    t.render_frames(0, args.frames)
    t.synchronize()
    rgba = t.read_output()
    if args.output.lower().endswith(".ppm"):
        image.write_ppm(args.output, rgba)
    else:
        image.write_png(args.output, rgba)
    if args.accum:
        image.save_accum(args.accum, t.read_accum(), args.frames, args.spp)
    print("%s: %dx%d, %d frames x %d spp, kernel %.3f ms" % (args.output, w, h, args.frames, args.spp, t.last_kernel_ms()))
    t.destroy()


if __name__ == "__main__":
    main()

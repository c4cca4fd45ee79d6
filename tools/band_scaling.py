import sys, os
sys.path.insert(0, os.getcwd())
import rtamd_loader
pkg = rtamd_loader.load()
for frames in (64, 1):
    t = pkg.RayTracer().create(1920, 1080, device_id=0)
    t.set_samples_per_pixel(16); t.set_max_depth(8); t.set_aperture(0.0)
    t.render_frames(0, frames); t.synchronize()
    for n in (4, 8, 16, 32, 64, 128, 256):
        best = 1e9
        for rep in range(3):
            t.render_shard(0, frames, 1, 800, 800 + n, write_output=False)
            best = min(best, t.kernel_ms(0))
        print("frames %d rows %4d: %.4f ms  %.5f ms/row  waves %d" % (frames, n, best, best / n, (1920 // (2 if frames == 64 else 8)) * (n // (4 if frames == 64 else 8))))
    t.destroy()

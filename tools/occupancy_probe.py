import sys, os
sys.path.insert(0, os.getcwd())
import rtamd_loader
pkg = rtamd_loader.load()
n = int(sys.argv[1])
t = pkg.RayTracer().create(1920, 1080, device_id=0)
t.set_samples_per_pixel(16); t.set_max_depth(50); t.set_aperture(0.05)
t.set_spheres(pkg.synthetic_scene(n)); t.set_camera([3.0, 1.5, 6.0], [-0.4, -0.2, -1.0], 6.0)
t.set_frame_overlap(False)
t.render_frames(0, 2); t.synchronize()
t.render_frames(0, 2)
print("spheres %d nodes %d pad %s: %.2f ms" % (n, t.get_scene_hierarchy()[0].shape[0], os.environ.get("RT_LDS_PAD", "0"), t.kernel_ms(0)))

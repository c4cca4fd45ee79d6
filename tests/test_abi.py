"""The C-ABI library loads on a machine without a GPU and exports every symbol include/rt_mi355x.h
declares; the host-side state machine (setters, clamps, reset flag, uniform block) is checked through a
handle created with RT_DEVICE_NONE against bytes from the reference's own RayTracer.cpp.  CPU only."""
import ctypes
import json
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "rt_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), "librt_mi355x.so does not export %s" % name
    assert sorted(pkg.ABI.keys()) == declared, "Python binding and header disagree"


def test_rccl_library_exports_every_declared_symbol(pkg):
    """The multi-GPU library (include/rt_mi355x_rccl.h) loads without a GPU and exports what its header declares."""
    from raytracing_amd import rccl
    text = open(os.path.join(ROOT, "include", "rt_mi355x_rccl.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))
    assert len(declared) >= 12
    lib = rccl.load_library()
    for name in declared:
        assert hasattr(lib, name), "librt_mi355x_rccl.so does not export %s" % name
    assert sorted(rccl.ABI.keys()) == declared, "Python binding and header disagree"
    # bad arguments are reported, not crashed on
    assert lib.rt_comm_destroy(None) != 0
    assert lib.rt_comm_world(None) == 0 and lib.rt_comm_rank(None) == -1
    out = ctypes.c_void_p()
    assert lib.rt_comm_create(bytes(128), 2, 5, 0, ctypes.byref(out)) != 0
    assert b"rank outside the world" in lib.rt_comm_last_error(None)


def test_struct_layouts(pkg):
    assert ctypes.sizeof(pkg.GPUSphere) == 48
    assert ctypes.sizeof(pkg.GPUParams) == 160
    assert pkg.GPUParams.frame_sample_depth_count.offset == 112
    assert pkg.GPUParams.resolution.offset == 128
    assert pkg.GPUParams.inv_resolution.offset == 144


def _words(hexlist):
    return np.array([int(h, 16) for h in hexlist], dtype=np.uint32)


def test_host_params_match_reference_bytes(pkg, golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "host_params.json")))
    for name, case in cases.items():
        argv = case["argv"]
        if argv[0] == "default":
            t = pkg.RayTracer().create(int(argv[1]), int(argv[2]), device_id=-1)
            t.set_samples_per_pixel(4)
            t.set_aperture(float(argv[3]))
        else:
            t = pkg.RayTracer().create(int(argv[0]), int(argv[1]), device_id=-1)
            t.set_aperture(float(argv[2]))
            t.set_fov(float(argv[3]))
            t.set_camera([float(a) for a in argv[5:8]], [float(a) for a in argv[8:11]], float(argv[4]))
        got = t.get_params_words(0)
        want = _words(case["params"])
        # the probe leaves the uvec4 zero (makeCameraParams); updateParams fills it
        assert np.array_equal(np.delete(got, range(28, 32)), np.delete(want, range(28, 32))), name
        assert np.array_equal(t.get_spheres().view(np.uint32).ravel(), _words(case["spheres"])), name
        t.destroy()


def test_setters_clamp_like_the_reference(pkg):
    t = pkg.RayTracer().create(64, 32, device_id=-1)
    p = t.get_params(7)
    assert list(p.frame_sample_depth_count) == [7, 4, 12, 5]      # RayTracer.h:92-93 defaults
    t.set_samples_per_pixel(0)
    t.set_max_depth(0)
    assert list(t.get_params(0).frame_sample_depth_count) == [0, 1, 1, 5]  # max(1, .)
    t.set_aperture(-3.0)
    assert t.get_params(0).origin_lens[3] == 0.0                    # max(0, .) * 0.5
    t.set_aperture(0.2)
    assert t.get_params(0).origin_lens[3] == np.float32(0.2) * np.float32(0.5)
    before = t.get_params_words(0)
    t.set_focus_distance(-1.0)                                      # ignored unless > 0
    t.set_focus_distance(0.0)
    assert np.array_equal(before, t.get_params_words(0))
    t.set_focus_distance(5.0)
    assert not np.array_equal(before, t.get_params_words(0))
    t.set_fov(1.0)                                                  # clamp to 5
    lo = t.get_params_words(0)
    t.set_fov(5.0)
    assert np.array_equal(lo, t.get_params_words(0))
    t.set_fov(500.0)                                                # clamp to 120
    hi = t.get_params_words(0)
    t.set_fov(120.0)
    assert np.array_equal(hi, t.get_params_words(0))
    t.resize(100, 50)
    assert t.extent == (100, 50)
    assert t.get_params(0).resolution[0] == 100.0 and t.get_params(0).inv_resolution[1] == np.float32(1.0) / np.float32(50.0)
    t.destroy()


def test_errors_are_status_codes_with_messages(pkg):
    lib = pkg.load_library()
    h = ctypes.c_void_p()
    assert lib.rt_create(-1, 0, 10, ctypes.byref(h)) == 1           # RT_ERR_BAD_ARGUMENT
    assert b"extent" in lib.rt_last_error(None)
    t = pkg.RayTracer().create(8, 8, device_id=-1)
    try:
        t.render(0)
        assert False, "render on a device-less handle must fail"
    except RuntimeError as e:
        assert "RT_DEVICE_NONE" in str(e)
    try:
        t.set_spheres(np.zeros((0, 12), dtype=np.float32))
        assert False
    except RuntimeError as e:
        assert "empty" in str(e)
    try:
        t.set_spheres(np.zeros(((1 << 20) + 1, 12), dtype=np.float32))
        assert False
    except RuntimeError as e:
        assert "1048576" in str(e)
    t.set_spheres(pkg.synthetic_scene(64))
    assert t.get_params(0).frame_sample_depth_count[3] == 64
    t.destroy()


def test_scene_generators(pkg, oracle):
    assert np.array_equal(pkg.default_scene().view(np.uint32), oracle.default_scene().view(np.uint32))
    a = pkg.synthetic_scene(300)
    b = pkg.synthetic_scene(300)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and a.shape == (300, 12)
    assert set(np.unique(a[:, 8])) <= {0.0, 1.0, 2.0}


def test_cpp_mirror_of_reference_class(pkg, golden_dir, tmp_path):
    """The header-only C++ RayTracer (same method names as src/rt/RayTracer.h:40-51) links against the
    C ABI and produces the reference's uniform block; render on a device-less handle throws."""
    import subprocess
    exe = str(tmp_path / "cpp_mirror_check")
    libdir = os.path.dirname(pkg.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "cpp_mirror_check.cpp"),
                           "-o", exe, "-L" + libdir, "-lrt_mi355x", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    cases = json.load(open(os.path.join(golden_dir, "host_params.json")))
    want = cases["default_1920x1080_ap0.05"]["params"]
    got = lines[0].split()
    assert got[:28] == want[:28] and got[32:] == want[32:]
    assert got[28:32] == ["00000000", "00000004", "0000000c", "00000005"]
    assert "RT_DEVICE_NONE" in lines[1]

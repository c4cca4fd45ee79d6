"""A caller written in the reference's host language: tests/cpp_frame_loop.cpp drives the C++ mirror of
class RayTracer (ray-tracing_amd/csrc/RayTracer.hpp) the way App::run does -- one render per frame, the
frame counter back to 0 after camera moves, setting changes and a resize -- in a process of its own on the
GPU.  The same calls are replayed here through the Python mirror, checked against the CPU oracle, and the
images of the two processes must agree word for word (compared through checksums).  Needs a real MI355X and
the program built beforehand (make -C tests, part of __graft_entry__.build)."""
import os
import subprocess

import numpy as np
import pytest

from test_gpu_parity import make_tracer, oracle_frames, same

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _word_sum(array):
    words = np.ascontiguousarray(array).view(np.uint32).ravel().astype(np.uint64)
    weights = np.arange(words.size, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
    return int((words * weights).sum(dtype=np.uint64))


def _stages_of(exe):
    run = subprocess.run([exe], capture_output=True, text=True, timeout=100)
    assert run.returncode == 0, run.stdout + run.stderr
    got = {}
    for line in run.stdout.strip().splitlines():
        stage, accum_sum, output_sum = line.split()
        got[stage] = (int(accum_sum, 16), int(output_sum, 16))
    return got


@pytest.fixture(scope="module")
def python_mirror_stages(pkg, oracle):
    """The calls of the C++ programs replayed through the Python mirror, every stage checked against the oracle."""
    scene = pkg.default_scene()
    t = pkg.RayTracer().create(320, 180, device_id=0)      # create-time defaults, as the C++ program has them
    t.set_samples_per_pixel(4)
    t.set_aperture(0.05)
    want = {}

    def stage(name, frames):
        ref = oracle_frames(oracle, t, scene, frames)
        assert same(t.read_accum(), ref["accum"])
        assert same(t.read_output(), ref["output"])
        want[name] = (_word_sum(t.read_accum()), _word_sum(t.read_output()))

    for f in range(6):
        t.render(f)
    stage("first_view", range(6))
    t.set_camera([10.0, 3.0, 6.0], [-0.8, -0.2, -0.5])
    for f in range(3):
        t.render(f)
    stage("camera_moved", range(3))
    t.set_max_depth(5)
    t.set_fov(35.0)
    t.set_focus_distance(9.0)
    for f in range(2):
        t.render(f)
    stage("settings_changed", range(2))
    t.resize(200, 120)
    t.render(0)
    t.render_frames(1, 4)
    stage("resized", range(5))
    t.destroy()
    return want


def test_cpp_caller_renders_what_the_python_mirror_renders(python_mirror_stages):
    exe = os.path.join(ROOT, "tests", "_bin", "cpp_frame_loop")
    assert os.path.isfile(exe), "tests/_bin/cpp_frame_loop is missing: run make -C tests (or __graft_entry__.build())"
    assert _stages_of(exe) == python_mirror_stages


def test_the_reference_class_bound_to_the_library_renders_the_same(python_mirror_stages):
    """tests/ref_class_frame_loop.cpp: the reference's own class RayTracer (its header, used where it lies) with
    ray-tracing_amd/csrc/RayTracer_mi355x.cpp in the place of the reference's RayTracer.cpp, driven like App::run.
    The program is built where the reference tree is mounted (make -C oracle ref) and travels as a binary."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_class_frame_loop")
    if not os.path.isfile(exe):
        pytest.skip("oracle/_ref/ref_class_frame_loop is built where the reference's headers are")
    assert _stages_of(exe) == python_mirror_stages

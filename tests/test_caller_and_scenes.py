"""The caller contract of App::run (frame counter ownership, reset rules) replayed through the C ABI on
a host-state-only handle, and the scene file round trips.  CPU only; the GPU half is in
test_gpu_parity.py::test_frame_loop_replay."""
import numpy as np
import pytest


class Recorder:
    """Wraps a device-less tracer: records render() calls instead of launching them."""

    def __init__(self, pkg, w=64, h=32):
        self.t = pkg.RayTracer().create(w, h, device_id=-1)
        self.calls = []

    def render(self, frame):
        self.calls.append(("render", frame, self.t.get_params_words(frame).copy()))

    def __getattr__(self, name):
        attr = getattr(self.t, name)
        if name.startswith("set_") or name == "resize":
            def wrapped(*a, **k):
                self.calls.append((name,) + tuple(a))
                return attr(*a, **k)
            return wrapped
        return attr


def test_frame_counter_rules(pkg):
    rec = Recorder(pkg)
    loop = pkg.FrameLoop(rec)
    assert [c[0] for c in rec.calls] == ["set_samples_per_pixel", "set_aperture"]   # App.cpp:155-156
    for _ in range(3):
        loop.step()
    assert loop.rendered == [0, 1, 2]
    loop.step(mouse=(10.0, -4.0))          # camera change: frame 0 is rendered in the same iteration
    assert loop.rendered[-1] == 0
    loop.step()
    assert loop.rendered[-1] == 1
    loop.step(sliders={"fov": 35.0})       # slider: applied after the frame; the next dispatch is frame 1
    assert loop.rendered[-1] == 2
    loop.step()
    assert loop.rendered[-1] == 1
    last = [c for c in rec.calls if c[0] == "render"][-1]
    assert last[2][28] == 1                # uvec4.x = frame index seen by the shader
    loop.step(resize=(80, 40))             # resize: nothing rendered, counter back to 0
    assert rec.t.extent == (80, 40)
    n = len(loop.rendered)
    loop.step()
    assert len(loop.rendered) == n + 1 and loop.rendered[-1] == 0


def test_camera_motion_matches_app_arithmetic(pkg):
    rec = Recorder(pkg)
    loop = pkg.FrameLoop(rec)
    # initial yaw/pitch reproduce the create-time aim
    d0 = loop.cam_dir.copy()
    loop.step(keys={"W"}, dt=0.5)
    moved = np.float32([13.0, 2.0, 3.0]) + d0 * (np.float32(5.0) * np.float32(0.5))
    assert np.allclose(loop.cam_pos, moved, rtol=0, atol=1e-6)
    cam_call = [c for c in rec.calls if c[0] == "set_camera"][-1]
    assert np.array_equal(np.float32(cam_call[1]), loop.cam_pos)
    loop.step(mouse=(0.0, 10000.0))        # pitch clamps at +89 degrees
    assert loop.pitch == np.float32(89.0)
    assert abs(float(np.linalg.norm(loop.cam_dir)) - 1.0) < 1e-6
    p = rec.t.get_params(0)
    assert p.origin_lens[0] == loop.cam_pos[0] and p.origin_lens[2] == loop.cam_pos[2]


def test_scene_files_round_trip(pkg, tmp_path):
    for scene in (pkg.default_scene(), pkg.synthetic_scene(50), pkg.random_scene()):
        for suffix in ("scene.json", "scene.spheres"):
            path = str(tmp_path / suffix)
            pkg.save_scene(path, scene)
            back = pkg.load_scene(path)
            assert back.shape == scene.shape
            assert np.array_equal(back.view(np.uint32), scene.view(np.uint32)), suffix
    r = pkg.random_scene()
    assert 400 < r.shape[0] <= 488 and np.array_equal(r.view(np.uint32), pkg.random_scene().view(np.uint32))
    with pytest.raises(ValueError):
        open(str(tmp_path / "bad.spheres"), "wb").write(b"\0" * 50)
        pkg.load_scene(str(tmp_path / "bad.spheres"))


def test_camera_arithmetic_matches_the_reference_glm(pkg, golden_dir):
    """tests/golden/caller_camera.json holds the reference's own camera arithmetic (src/core/App.cpp:180-183,
    :299-362) evaluated with its vendored GLM on a scripted sequence of mouse deltas and key presses;
    FrameLoop must reproduce position, direction, yaw and pitch bit for bit after every step."""
    import json
    import os
    fixture = json.load(open(os.path.join(golden_dir, "caller_camera.json")))
    rec = Recorder(pkg)
    loop = pkg.FrameLoop(rec)

    def state():
        vals = list(loop.cam_pos) + list(loop.cam_dir) + [loop.yaw, loop.pitch]
        return ["%08x" % int(np.float32(v).view(np.uint32)) for v in vals]

    assert state() == fixture["init"]
    names = {"W": "W", "S": "S", "A": "A", "D": "D", "U": "SPACE", "P": "SHIFT"}
    for i, ev in enumerate(fixture["steps"]):
        loop.step(dt=ev["dt"], mouse=(ev["dx"], ev["dy"]), keys={names[k] for k in ev["keys"]})
        assert state() == ev["state"], "step %d" % i
        if ev["keys"] or ev["dx"] or ev["dy"]:
            assert loop.rendered[-1] == 0      # any camera change restarts the accumulation (:364-368)

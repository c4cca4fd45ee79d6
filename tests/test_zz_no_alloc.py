"""rt_render_frames_static -- the entry point that allocates nothing (its per-frame sums and unit counter
live in the library's file-scope device array) -- against rt_render_frames, through the C ABI: same
accumulator floats, RNG words and rgba8 bytes, on two calls in a row that share the array, the second
with a smaller input.  Needs a real MI355X."""
import numpy as np
import pytest

from test_gpu_parity import make_tracer, same

pytestmark = pytest.mark.gpu


def _pair(pkg, w, h, spp, depth, scene=None):
    a = make_tracer(pkg, w, h, spp, depth, 0.05, scene=scene)
    b = make_tracer(pkg, w, h, spp, depth, 0.05, scene=scene)
    a.enable_rng_capture(True)
    b.enable_rng_capture(True)
    return a, b


def _same_images(a, b):
    assert same(a.read_rng_state(), b.read_rng_state())
    assert same(a.read_accum(), b.read_accum())
    assert same(a.read_output(), b.read_output())


def test_static_scratch_entry_point_matches_render_frames(pkg):
    lib = pkg.load_library()
    room = lib.rt_static_scratch_bytes()
    assert room == 64 << 20
    # first call: 640x360, 40 frames = 147 MB of per-frame sums -> three groups through the 64 MiB array
    ref, got = _pair(pkg, 640, 360, 2, 8)
    ref.render_frames(0, 40)
    got.render_frames_static(0, 40)
    _same_images(ref, got)
    # second call, smaller input, hierarchy scene (the unit counter lives in the array too); the array
    # still holds the first call's sums
    scene = pkg.synthetic_scene(200)
    ref2, got2 = _pair(pkg, 200, 120, 3, 12, scene=scene)
    ref2.render_frames(0, 5)
    got2.render_frames_static(0, 5)
    _same_images(ref2, got2)
    # continuing an accumulation, single frames, and the first handle again after the second used the array
    ref2.render_frames(5, 1)
    got2.render_frames_static(5, 1)
    _same_images(ref2, got2)
    ref.render_frames(40, 7)
    got.render_frames_static(40, 7)
    _same_images(ref, got)
    for t in (ref, got, ref2, got2):
        t.destroy()


def test_static_scratch_rejects_images_it_cannot_hold(pkg):
    # two frames of 2048x1100 need 72 MB: more than the array; one frame needs no per-frame sums at all
    t = make_tracer(pkg, 2048, 1100, 1, 4)
    with pytest.raises(RuntimeError, match="static scratch holds"):
        t.render_frames_static(0, 2)
    t.render_frames_static(0, 1)
    r = make_tracer(pkg, 2048, 1100, 1, 4)
    r.render_frames(0, 1)
    assert same(t.read_accum(), r.read_accum())
    # 1920x1080 is the largest standard size of which two frames fit
    u = make_tracer(pkg, 1920, 1080, 1, 4)
    u.render_frames_static(0, 3)
    v = make_tracer(pkg, 1920, 1080, 1, 4)
    v.render_frames(0, 3)
    assert same(u.read_accum(), v.read_accum())
    for x in (t, r, u, v):
        x.destroy()


@pytest.mark.parametrize("scene_size", [5, 200])
def test_static_scratch_entry_point_inside_a_stream_capture(pkg, scene_size):
    """What the entry point is for: a stream capture forbids allocation and synchronisation.  One call is
    captured into a HIP graph (nothing runs during the capture), the graph is replayed twice, and each replay
    leaves the image an ordinary rt_render_frames call leaves; afterwards the handle works outside the graph,
    a scene change included (which waits for the handle's kernels and cannot wait for a captured event)."""
    import torch
    w, h, frames = 320, 200, 5
    scene = pkg.default_scene() if scene_size == 5 else pkg.synthetic_scene(scene_size)
    ref = make_tracer(pkg, w, h, 2, 6, 0.05, scene=scene)
    ref.render_frames(0, frames)
    want_accum, want_output = ref.read_accum(), ref.read_output()
    stream = torch.cuda.Stream()
    t = make_tracer(pkg, w, h, 2, 6, 0.05, scene=scene)
    t.set_stream(stream.cuda_stream)
    marker = np.full((h, w, 4), 7.0, dtype=np.float32)
    t.write_accum(marker)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream):
        t.render_frames_static(0, frames)
    assert same(t.read_accum(), marker)             # captured, not run
    for _ in range(2):
        graph.replay()
        torch.cuda.synchronize()
        assert same(t.read_accum(), want_accum)
        assert same(t.read_output(), want_output)
    t.set_spheres(scene)
    t.render_frames(0, frames)
    assert same(t.read_accum(), want_accum)
    del graph
    t.destroy()
    ref.destroy()


@pytest.mark.parametrize("scene_size", [5, 200])
def test_plain_render_inside_a_stream_capture(pkg, scene_size):
    """rt_render with the library's defaults (launch overlap on) on a stream that is being captured: the
    library sees the capture and keeps the whole call on the caller's stream, so it becomes part of the
    graph instead of running at once on the library's own streams.  One frame needs no per-frame sums, so
    nothing is allocated either.  Replays leave the image of an ordinary call."""
    import torch
    w, h = 320, 200
    scene = pkg.default_scene() if scene_size == 5 else pkg.synthetic_scene(scene_size)
    ref = make_tracer(pkg, w, h, 2, 6, 0.05, scene=scene)
    ref.render(0)
    want_accum, want_output = ref.read_accum(), ref.read_output()
    ref.render(1)
    want_accum2, want_output2 = ref.read_accum(), ref.read_output()
    stream = torch.cuda.Stream()
    t = make_tracer(pkg, w, h, 2, 6, 0.05, scene=scene)
    t.set_stream(stream.cuda_stream)
    t.render(0)                      # an ordinary, overlapped call first: the handle owns the device's overlap
    t.synchronize()
    assert same(t.read_accum(), want_accum)
    marker = np.full((h, w, 4), 7.0, dtype=np.float32)
    t.write_accum(marker)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream):
        t.render(0)                  # frame 0 clears, then traces: both captured
    assert same(t.read_accum(), marker)             # captured, not run
    for _ in range(2):
        graph.replay()
        torch.cuda.synchronize()
        assert same(t.read_accum(), want_accum)
        assert same(t.read_output(), want_output)
    # and the handle goes on outside the graph, overlapped again
    t.render(1)
    t.synchronize()
    assert same(t.read_accum(), want_accum2)
    assert same(t.read_output(), want_output2)
    del graph
    t.destroy()
    ref.destroy()

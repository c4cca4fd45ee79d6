"""Parity of the HIP path (through the C ABI) with the CPU oracle, bit for bit: RNG words, accumulator
floats, rgba8 bytes.  Needs a real MI355X."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def same(a, b):
    """Exact equality with a short failure message (arrays here are too large to print)."""
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    if a.dtype == np.float32:
        a, b = a.view(np.uint32), b.view(np.uint32)
    if a.shape != b.shape:
        pytest.fail("shape %s != %s" % (a.shape, b.shape), pytrace=False)
    bad = int(np.count_nonzero(a != b))
    if bad:
        idx = np.argwhere(a != b)[0]
        pytest.fail("%d of %d elements differ, first at %s: %s vs %s" % (bad, a.size, idx.tolist(), a[tuple(idx)], b[tuple(idx)]),
                    pytrace=False)
    return True


def oracle_frames(oracle, tracer, scene, frames, row_begin=0, row_end=None, accum=None, want_rng=False):
    """Runs the oracle for the given frame indices with the tracer's own uniform block."""
    out = None
    for f in frames:
        words = tracer.get_params_words(f)
        out = oracle.render(words, scene, accum=accum, y0=row_begin, y1=row_end, want_rng=want_rng)
        accum = out["accum"]
    return out


def make_tracer(pkg, w, h, spp=2, depth=8, aperture=0.05, scene=None):
    t = pkg.RayTracer().create(w, h, device_id=0)
    t.set_samples_per_pixel(spp)
    t.set_max_depth(depth)
    t.set_aperture(aperture)
    if scene is not None:
        t.set_spheres(scene)
    return t


def test_rng_words_match_known_answers(pkg, oracle, golden_dir):
    import json
    import os
    kat = json.load(open(os.path.join(golden_dir, "rng_kat.json")))
    xyz = np.array([r["xyz"] for r in kat], dtype=np.uint32)
    got = pkg.probe_rng(xyz, 8)
    for i, rec in enumerate(kat):
        assert int(got[i, 0]) == rec["hash"]
        assert [int(v) for v in got[i, 1:]] == rec["lcg"]
    # and a random batch against the oracle
    rng = np.random.default_rng(3)
    xyz = rng.integers(0, 2**32, size=(4096, 3), dtype=np.uint64).astype(np.uint32)
    got = pkg.probe_rng(xyz, 4)
    for i in range(0, 4096, 97):
        h = oracle.hash3(int(xyz[i, 0]), int(xyz[i, 1]), int(xyz[i, 2]))
        assert int(got[i, 0]) == h
        assert [int(v) for v in got[i, 1:]] == oracle.lcg_sequence(h, 4)


def test_pow_bits_match_oracle(pkg, oracle):
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.random(20000, dtype=np.float32),
                        np.float32([0.0, 1.0, 0.5, 1e-6, 1e-30, 1e-42, 0.99999994, 2.0, 100.0])])
    for yv in (5.0, float(np.float32(1.0 / 2.2))):
        y = np.full_like(x, yv)
        got = pkg.probe_pow(x, y)
        want = np.array([oracle.lib().rto_powf(float(a), float(b)) for a, b in zip(x, y)], dtype=np.float32)
        assert same(got, want)


def _same_or_both_nan(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    bad = (a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))
    if bad.any():
        i = int(np.argmax(bad))
        pytest.fail("%d of %d differ, first at %d: %r (0x%08x) vs %r (0x%08x)" % (
            int(bad.sum()), a.size, i, a[i], a.view(np.uint32)[i], b[i], b.view(np.uint32)[i]), pytrace=False)
    return True


def test_cheap_divide_and_sqrt_are_correctly_rounded(pkg):
    # The tracer's divide/sqrt helpers skip the range-scaling steps of the IEEE sequences when the
    # operands allow.  They must give the correctly rounded result (numpy float32 = IEEE) for every
    # input: random bit patterns over the whole exponent range, the ranges the tracer lives in, and
    # values around the range limits where the helpers switch to the full sequence.
    rng = np.random.default_rng(11)
    n = 4_000_000
    old = np.seterr(all="ignore")
    try:
        def patterns(m):
            return rng.integers(0, 2**32, size=m, dtype=np.uint64).astype(np.uint32).view(np.float32)

        def typical(m, lo, hi):
            return (np.exp(rng.uniform(np.log(lo), np.log(hi), size=m)) * rng.choice([-1.0, 1.0], size=m)).astype(np.float32)

        edges = np.float32([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 2.0**-20, 2.0**20, 2.0**-90, 2.0**100,
                            2.0**-40, 2.0**40, 1e-45, 1e-38, 3e38, 2.0**-126, 2.0**-127, 0.99999994, 1.0000001,
                            np.nextafter(np.float32(2.0**-20), np.float32(0)), np.nextafter(np.float32(2.0**20), np.float32(np.inf)),
                            np.nextafter(np.float32(2.0**-90), np.float32(0)), np.nextafter(np.float32(2.0**100), np.float32(0)),
                            np.nextafter(np.float32(2.0**-40), np.float32(0)), np.nextafter(np.float32(2.0**40), np.float32(np.inf))])
        ea, eb = np.meshgrid(edges, edges)
        # division
        for a, b in [(patterns(n), patterns(n)),
                     (typical(n, 1e-6, 1e6), typical(n, 1e-3, 1e3)),
                     (typical(n, 1e-30, 1e30), np.abs(typical(n, 0.5, 2.0))),
                     (ea.ravel(), eb.ravel())]:
            cheap, full = pkg.probe_arith(0, a, b)
            want = a / b
            assert _same_or_both_nan(full, want)
            assert _same_or_both_nan(cheap, want)
        # sqrt and 1/sqrt
        for a in [np.abs(patterns(n)), patterns(n), np.abs(typical(n, 1e-12, 1e12)), np.abs(typical(n, 0.25, 4.0)), edges]:
            cheap, full = pkg.probe_arith(1, a)
            want = np.sqrt(a)
            assert _same_or_both_nan(full, want)
            assert _same_or_both_nan(cheap, want)
            cheap, full = pkg.probe_arith(2, a)
            want = np.float32(1.0) / np.sqrt(a)
            assert _same_or_both_nan(full, want)
            assert _same_or_both_nan(cheap, want)
    finally:
        np.seterr(**old)


def test_inversesqrt_over_its_whole_short_range(pkg):
    """Every 16th binary32 value of [2^-40, 2^40] (tools/exhaustive_inversesqrt.py runs all of them): the
    kernel's inversesqrt against the full IEEE sqrt and division sequences on the device."""
    lo = np.float32(2.0 ** -40).view(np.uint32).item()
    hi = np.float32(2.0 ** 40).view(np.uint32).item()
    for start in range(lo, hi + 1, 1 << 28):
        bits = np.arange(start, min(start + (1 << 28), hi + 1), 16, dtype=np.uint32)
        cheap, full = pkg.probe_arith(2, bits.view(np.float32))
        assert np.array_equal(cheap.view(np.uint32), full.view(np.uint32))


def test_hit_sphere_vectors(pkg, oracle):
    rng = np.random.default_rng(7)
    n = 6000
    scene = pkg.default_scene()
    spheres = scene[rng.integers(0, 5, size=n)]
    c = spheres[:, :3]
    r = spheres[:, 3:4]
    o = rng.normal(size=(n, 3)).astype(np.float32) * np.float32(3.0) + np.float32([0, 1, 0])
    # aim at the sphere with a spread of about two radii, so hits, grazes and misses all occur
    d = (c - o) + rng.normal(size=(n, 3)).astype(np.float32) * r * np.float32(0.7)
    d = d.astype(np.float32)
    d[: n // 2] /= np.linalg.norm(d[: n // 2], axis=1, keepdims=True).astype(np.float32)
    # a third of the rays start on the sphere surface in a random direction (the far-root branch)
    k = n // 3
    sdir = rng.normal(size=(k, 3)).astype(np.float32)
    sdir /= np.linalg.norm(sdir, axis=1, keepdims=True).astype(np.float32)
    o[:k] = (c[:k] + r[:k] * sdir).astype(np.float32)
    d[:k] = rng.normal(size=(k, 3)).astype(np.float32)
    tmax = np.where(rng.random(n) < 0.5, np.float32(1e20), rng.random(n).astype(np.float32) * np.float32(20.0))
    rays = np.zeros((n, 8), dtype=np.float32)
    rays[:, 0:3] = o
    rays[:, 3] = 0.001
    rays[:, 4:7] = d
    rays[:, 7] = tmax
    out, front = pkg.probe_hit_sphere(spheres, rays)
    hits = 0
    for i in range(n):
        ref = oracle.hit_sphere(spheres[i], rays[i, 0:3], rays[i, 4:7], 0.001, float(tmax[i]))
        if ref is None:
            assert out[i, 0] == 0.0
            continue
        hits += 1
        assert out[i, 0] == 1.0
        want = np.float32([ref["t"]] + ref["p"] + ref["n"])
        assert same(out[i, 1:8], want)
        assert bool(front[i]) == ref["front"]
    assert hits > n // 4


@pytest.mark.parametrize("w,h,spp,depth,aperture", [
    (64, 64, 1, 8, 0.05),
    (200, 120, 4, 8, 0.0),      # DoF off: the lens draws are still consumed
    (67, 45, 3, 1, 0.05),       # ragged extent, one bounce
    (1, 1, 16, 12, 0.2),
    (9, 130, 2, 50, 0.05),
])
def test_single_frame_bit_exact(pkg, oracle, w, h, spp, depth, aperture):
    t = make_tracer(pkg, w, h, spp, depth, aperture)
    t.enable_rng_capture(True)
    t.render(0)
    got_acc, got_out, got_rng = t.read_accum(), t.read_output(), t.read_rng_state()
    ref = oracle_frames(oracle, t, pkg.default_scene(), [0], want_rng=True)
    assert same(got_rng, ref["rng"])
    assert same(got_acc, ref["accum"])
    assert same(got_out, ref["output"])
    t.destroy()


def test_accumulate_frames_sequential_and_batched(pkg, oracle):
    w, h = 160, 90
    scene = pkg.default_scene()
    seq = make_tracer(pkg, w, h, 2, 8)
    for f in range(5):
        seq.render(f)
    bat = make_tracer(pkg, w, h, 2, 8)
    bat.enable_rng_capture(True)
    bat.render_frames(0, 5)
    ref = oracle_frames(oracle, seq, scene, range(5), want_rng=True)
    assert same(seq.read_accum(), ref["accum"])
    assert same(bat.read_accum(), ref["accum"])
    assert same(bat.read_rng_state(), ref["rng"])
    assert same(seq.read_output(), ref["output"])
    assert same(bat.read_output(), ref["output"])
    # continuing the accumulation from frame 5 without a reset
    bat.render_frames(5, 2)
    ref2 = oracle_frames(oracle, seq, scene, [5, 6], accum=ref["accum"].copy())
    assert same(bat.read_accum(), ref2["accum"])
    seq.destroy()
    bat.destroy()


def test_reset_on_change_semantics(pkg, oracle):
    # RayTracer::render clears iff the flag is up or frame_index == 0 (src/rt/RayTracer.cpp:564)
    w, h = 48, 32
    scene = pkg.default_scene()
    t = make_tracer(pkg, w, h, 1, 4)
    t.render(0)
    t.render(1)
    two = t.read_accum()
    t.render(0)                      # frame 0 clears
    one = t.read_accum()
    ref0 = oracle_frames(oracle, t, scene, [0])
    assert same(one, ref0["accum"])
    assert not np.array_equal(bits(one), bits(two))
    t.set_fov(30.0)                  # any setter raises the flag ...
    t.render(3)                      # ... so frame 3 starts from a cleared accumulator (the App.cpp:424-448 quirk)
    ref3 = oracle_frames(oracle, t, scene, [3])
    assert same(t.read_accum(), ref3["accum"])
    assert same(t.read_output(), ref3["output"])
    t.render(4)                      # flag is down again: accumulates
    ref34 = oracle_frames(oracle, t, scene, [4], accum=ref3["accum"].copy())
    assert same(t.read_accum(), ref34["accum"])
    t.set_focus_distance(-1.0)       # ignored: does not raise the flag
    t.render(5)
    ref345 = oracle_frames(oracle, t, scene, [5], accum=ref34["accum"].copy())
    assert same(t.read_accum(), ref345["accum"])
    t.destroy()


def test_resize_reallocates_and_resets(pkg, oracle):
    t = make_tracer(pkg, 40, 40, 1, 4)
    t.render(0)
    t.resize(72, 24)
    assert t.extent == (72, 24)
    t.render(2)
    ref = oracle_frames(oracle, t, pkg.default_scene(), [2])
    assert same(t.read_accum(), ref["accum"])
    t.destroy()


def test_row_band_shards_equal_whole_image(pkg, oracle):
    w, h = 96, 70
    t = make_tracer(pkg, w, h, 2, 8)
    t.render_frames(0, 3)
    whole = t.read_accum()
    whole_out = t.read_output()
    s = make_tracer(pkg, w, h, 2, 8)
    for (a, b) in [(0, 9), (9, 41), (41, 70)]:
        s.render_shard(0, 3, 1, a, b)
    assert same(s.read_accum(), whole)
    assert same(s.read_output(), whole_out)
    t.destroy()
    s.destroy()


def test_frame_stride_shards(pkg, oracle):
    # frame-index split: two "ranks" take even and odd frames; the RNG streams stay those of the
    # single-GPU run, the sum differs from the sequential one only by float association
    w, h = 80, 60
    scene = pkg.default_scene()
    even = make_tracer(pkg, w, h, 2, 8)
    odd = make_tracer(pkg, w, h, 2, 8)
    even.render_shard(0, 4, 2, 0, h, write_output=False)
    odd.render_shard(1, 4, 2, 0, h, write_output=False)
    ref_even = oracle_frames(oracle, even, scene, [0, 2, 4, 6])
    ref_odd = oracle_frames(oracle, even, scene, [1, 3, 5, 7])
    assert same(even.read_accum(), ref_even["accum"])
    assert same(odd.read_accum(), ref_odd["accum"])
    seq = oracle_frames(oracle, even, scene, range(8))
    summed = ref_even["accum"][..., :3] + ref_odd["accum"][..., :3]
    lum = seq["accum"][..., :3].mean()
    rmse = np.sqrt(np.mean((summed - seq["accum"][..., :3]) ** 2))
    assert rmse <= 1e-3 * lum
    # resolve after the "reduction": write the sum back and run the display tail alone
    total = np.concatenate([summed, np.ones((h, w, 1), dtype=np.float32)], axis=2)
    even.write_accum(total)
    even.resolve_output(8 * 2)
    # the display tail of the summed image, bit for bit; the sequential image is within one 8-bit step of it
    assert same(even.read_output(), oracle.resolve(total, 8 * 2))
    diff = np.abs(even.read_output().astype(np.int32) - seq["output"].astype(np.int32))
    assert diff.max() <= 1
    even.destroy()
    odd.destroy()


@pytest.mark.parametrize("count,hierarchy", [(2, True), (23, True), (24, True), (24, False), (63, True), (64, True), (64, False), (1024, True), (1024, False),
                                             (1500, True), (1500, False), (4096, True), (4096, False),
                                             (24, 2), (64, 2), (1024, 2), (4096, 2), (65, True), (257, True), (4097, True)])
def test_custom_scenes(pkg, oracle, count, hierarchy):
    # with the hierarchy off, 1500 and 4096 spheres exercise the geometry-only LDS layout of the loop
    # over all spheres (materials fetched from global memory); with it on, scenes of 24 spheres or
    # more are walked through the bounding-sphere hierarchy
    scene = pkg.synthetic_scene(count)
    w, h = (48, 32) if count > 1024 else (80, 48)
    t = make_tracer(pkg, w, h, 2, 6, 0.05, scene=scene)
    t.set_scene_hierarchy(hierarchy)
    t.set_camera([3.0, 1.5, 6.0], [-0.4, -0.2, -1.0], 6.0)
    t.enable_rng_capture(True)
    t.render_frames(0, 2)
    ref = oracle_frames(oracle, t, scene, [0, 1], want_rng=True)
    assert same(t.read_rng_state(), ref["rng"])
    assert same(t.read_accum(), ref["accum"])
    assert same(t.read_output(), ref["output"])
    t.destroy()


@pytest.mark.parametrize("w,h,frames", [(37, 23, 64), (45, 31, 20), (19, 9, 40), (8, 4, 64), (3, 2, 7)])
def test_unit_geometries_with_ragged_tiles(pkg, oracle, w, h, frames):
    """The pixel tile of a unit shrinks as the launch gets more frames (8x8 for one frame down to 2x4
    for 64); image sizes that are no multiple of any of them must still cover every pixel once."""
    scene = pkg.default_scene()
    t = make_tracer(pkg, w, h, 2, 6, 0.05)
    t.enable_rng_capture(True)
    t.render_frames(0, frames)
    ref = oracle_frames(oracle, t, scene, range(frames), want_rng=True)
    assert same(t.read_rng_state(), ref["rng"])
    assert same(t.read_accum(), ref["accum"])
    assert same(t.read_output(), ref["output"])
    t.destroy()


@pytest.mark.parametrize("pos", [(13.0, 2.0, 3.0), (0.0, 2.0, 9.0), (-0.0, 2.0, 9.0), (5.0, 0.0, -0.0), (0.0, 0.5, 0.0)])
def test_pinhole_camera_positions(pkg, oracle, pos):
    """Aperture 0: the lens offset is +-0.  The kernel skips it only when no component of the camera
    position is zero; zero (and minus zero) components must give the oracle's bits as well."""
    scene = pkg.default_scene()
    t = make_tracer(pkg, 96, 54, 3, 8, 0.0)
    t.set_camera(list(pos), [0.1 - pos[0], 1.0 - pos[1], 0.2 - pos[2]], 8.0)
    t.enable_rng_capture(True)
    t.render_frames(0, 2)
    ref = oracle_frames(oracle, t, scene, [0, 1], want_rng=True)
    assert same(t.read_rng_state(), ref["rng"])
    assert same(t.read_accum(), ref["accum"])
    assert same(t.read_output(), ref["output"])
    t.destroy()


def test_full_1080p_frame_bit_exact(pkg, oracle):
    # the benchmark configuration (1920x1080, 16 spp, depth 8, DoF off), one frame
    t = make_tracer(pkg, 1920, 1080, 16, 8, 0.0)
    t.enable_rng_capture(True)
    t.render(0)
    ref = oracle_frames(oracle, t, pkg.default_scene(), [0], want_rng=True)
    assert same(t.read_rng_state(), ref["rng"])
    assert same(t.read_accum(), ref["accum"])
    assert same(t.read_output(), ref["output"])
    t.destroy()


def test_benchmark_batch_equals_frame_by_frame_at_full_size(pkg):
    """The benchmark workload itself (1920x1080, 16 spp, depth 8, DoF off, 64 frames in one launch: 2 x 4
    pixel tiles, per-frame sums folded in order) against 64 single-frame launches (8 x 8 tiles, direct
    accumulation), whose first frame test_full_1080p_frame_bit_exact pins to the oracle; and against two
    row shards of the same batch."""
    batch = make_tracer(pkg, 1920, 1080, 16, 8, 0.0)
    batch.enable_rng_capture(True)
    batch.render_frames(0, 64)
    single = make_tracer(pkg, 1920, 1080, 16, 8, 0.0)
    single.enable_rng_capture(True)
    for f in range(64):
        single.render(f)
    assert same(batch.read_rng_state(), single.read_rng_state())
    assert same(batch.read_accum(), single.read_accum())
    assert same(batch.read_output(), single.read_output())
    single.destroy()
    shards = make_tracer(pkg, 1920, 1080, 16, 8, 0.0)
    shards.render_shard(0, 64, 1, 0, 541, write_output=False)
    shards.render_shard(0, 64, 1, 541, 1080, write_output=False)
    assert same(batch.read_accum(), shards.read_accum())
    shards.destroy()
    batch.destroy()


def test_converged_radiance_properties(pkg, oracle):
    # size-independent checks at 4096 spp on a small image: alpha is 1, radiance is finite and
    # non-negative, and the mean luminance agrees with an independent 256-spp oracle render of
    # other frame indices (different random streams) within Monte-Carlo noise
    w, h = 64, 36
    t = make_tracer(pkg, w, h, 64, 8)
    t.render_frames(0, 64)
    acc = t.read_accum()
    assert np.all(acc[..., 3] == 1.0) and np.all(np.isfinite(acc)) and np.all(acc[..., :3] >= 0.0)
    mean_gpu = acc[..., :3].mean() / 4096.0
    ref = oracle_frames(oracle, t, pkg.default_scene(), range(100, 104))
    mean_ref = ref["accum"][..., :3].mean() / 256.0
    assert abs(mean_gpu - mean_ref) <= 0.02 * mean_ref
    t.destroy()


def _replay(pkg, oracle, script, w, h):
    t = pkg.RayTracer().create(w, h, device_id=0)

    class Tap:
        """Intercepts render() to run the oracle with the uniform block of that very dispatch."""

        def __init__(self):
            self.accum = np.zeros((h, w, 4), dtype=np.float32)
            self.flag = True      # rt_create raises the reset flag
            self.out = None

        def render(self, frame):
            if self.flag or frame == 0:
                self.accum[...] = 0.0
            self.flag = False
            self.out = oracle.render(t.get_params_words(frame), pkg.default_scene(), accum=self.accum)
            t.render(frame)

        def __getattr__(self, name):
            attr = getattr(t, name)
            if name.startswith("set_") or name == "resize":
                def wrapped(*a, **k):
                    if name != "set_focus_distance" or a[0] > 0:
                        self.flag = True
                    return attr(*a, **k)
                return wrapped
            return attr

    tap = Tap()
    loop = pkg.FrameLoop(tap)
    tap.set_max_depth(6)
    for ev in script:
        loop.step(**ev)
    got_acc, got_out = t.read_accum(), t.read_output()
    t.destroy()
    return loop, tap, got_acc, got_out


def test_frame_loop_replay_matches_oracle(pkg, oracle):
    script = [dict(), dict(), dict(mouse=(25.0, 5.0)), dict(), dict(sliders={"samples": 2}), dict(),
              dict(keys={"W", "D"}, dt=0.1), dict(), dict(sliders={"aperture": 0.0}), dict(), dict()]
    loop, tap, got_acc, got_out = _replay(pkg, oracle, script, 72, 40)
    assert loop.rendered == [0, 1, 0, 1, 2, 1, 0, 1, 2, 1, 2]
    assert same(got_acc, tap.accum)
    assert same(got_out, tap.out["output"])


@pytest.mark.parametrize("scene_size", [5, 200])
def test_overlapped_single_frame_calls(pkg, oracle, scene_size):
    """Consecutive rt_render calls are traced on two internal streams and folded in call order: same bits
    as with the overlap switched off and as the oracle, across resets, a shard call and a batch in between
    (the 200-sphere scene goes through the hierarchy kernel, whose overlapping launches need a unit counter
    each)."""
    w, h = 192, 108
    scene = pkg.default_scene() if scene_size == 5 else pkg.synthetic_scene(scene_size)
    a = make_tracer(pkg, w, h, 3, 8, 0.05, scene=scene)
    b = make_tracer(pkg, w, h, 3, 8, 0.05, scene=scene)
    b.set_frame_overlap(False)
    for t in (a, b):
        for f in range(9):
            t.render(f)
    ref = oracle_frames(oracle, a, scene, range(9))
    assert same(a.read_accum(), ref["accum"])
    assert same(a.read_output(), ref["output"])
    assert same(b.read_accum(), ref["accum"])
    for t in (a, b):
        t.render_frames(9, 3)           # a batch on the caller's stream between overlapped calls
        t.render(12)
        t.render_shard(13, 1, 1, 0, 50)
        t.render_shard(13, 1, 1, 50, h)
        t.render(14)
    ref = oracle_frames(oracle, a, scene, range(9, 15), accum=ref["accum"])
    assert same(a.read_accum(), ref["accum"])
    assert same(a.read_output(), ref["output"])
    assert same(b.read_accum(), a.read_accum())
    a.set_fov(33.0)                     # reset in the middle of a run of overlapped calls
    for f in (4, 5, 6):
        a.render(f)
    ref = oracle_frames(oracle, a, scene, (4, 5, 6))
    assert same(a.read_accum(), ref["accum"])
    assert same(a.read_output(), ref["output"])
    a.destroy()
    b.destroy()


def test_large_batch_then_small_overlapped_batches_share_a_buffer(pkg, oracle):
    """A large multi-frame launch (caller's stream) followed at once by small multi-frame launches (traced on the
    library's own streams into the same per-frame-sum buffer): the small ones must wait for the large one's fold."""
    w, h = 1920, 1080
    scene = pkg.default_scene()
    t = make_tracer(pkg, w, h, 1, 4, 0.0)
    ref = make_tracer(pkg, w, h, 1, 4, 0.0)
    ref.set_frame_overlap(False)
    for tr in (t, ref):
        tr.render_frames(0, 24)                    # 50 M items: large, not overlapped
        for k in range(6):
            tr.render_shard(24 + 2 * k, 2, 1, 0, 64)   # 245 K items each: small, overlapped when allowed
        tr.render_frames(36, 24)
    assert same(t.read_accum(), ref.read_accum())
    assert same(t.read_output(), ref.read_output())
    band = oracle_frames(oracle, t, scene, [0], row_begin=0, row_end=8)
    t.render_shard(0, 1, 1, 0, 8)
    assert same(t.read_accum()[0:8], band["accum"][0:8])
    t.destroy()
    ref.destroy()


@pytest.mark.parametrize("caller_priority", [0, -1])
def test_overlapped_calls_beside_application_streams(pkg, oracle, caller_priority):
    """The library traces overlapped calls on two streams of the highest priority.  An application that
    keeps high-priority streams of its own busy, and whose own stream is of normal or of high priority, gets
    the same image and no stall (the suite's time limit ends a hang)."""
    import torch
    w, h, frames = 320, 180, 24
    scene = pkg.default_scene()
    caller = torch.cuda.Stream(priority=caller_priority)
    busy = [torch.cuda.Stream(priority=-1) for _ in range(4)]
    buffers = [torch.zeros(1 << 18, device="cuda") for _ in busy]
    t = make_tracer(pkg, w, h, 3, 8, 0.05)
    t.set_stream(caller.cuda_stream)
    for f in range(frames):
        for s, b in zip(busy, buffers):
            with torch.cuda.stream(s):
                b.add_(1.0)
        t.render(f)
    t.synchronize()
    torch.cuda.synchronize()
    ref = oracle_frames(oracle, t, scene, range(frames))
    assert same(t.read_accum(), ref["accum"])
    assert same(t.read_output(), ref["output"])
    assert all(float(b[0]) == frames for b in buffers)
    t.destroy()


def test_overlapped_calls_survive_stream_and_tracer_changes(pkg, oracle):
    """Sequences that once stalled the launch overlap for good (the suite's time limit ends a hang): a tracer
    used on one PyTorch stream and destroyed, then another on a second stream; several tracers at once, each on
    a stream of its own (one of them overlaps); one tracer whose caller alternates between two streams."""
    import torch
    w, h, frames = 256, 144, 20
    scene = pkg.default_scene()
    streams = [torch.cuda.Stream() for _ in range(4)]

    def fresh(stream):
        t = make_tracer(pkg, w, h, 3, 8, 0.05)
        t.set_stream(stream.cuda_stream)
        return t

    first = fresh(streams[0])
    for f in range(frames):
        first.render(f)
    want = oracle_frames(oracle, first, scene, range(frames))["accum"]
    assert same(first.read_accum(), want)
    first.destroy()
    second = fresh(streams[1])
    for f in range(frames):
        second.render(f)
    assert same(second.read_accum(), want)
    second.destroy()

    crowd = [fresh(s) for s in streams]
    for f in range(frames):
        for t in crowd:
            t.render(f)
    for t in crowd:
        assert same(t.read_accum(), want)
    crowd[0].destroy()                   # the overlap passes to whoever asks next
    for f in range(frames):
        for t in crowd[1:]:
            t.render(f)
    for t in crowd[1:]:
        assert same(t.read_accum(), want)
        t.destroy()

    t = fresh(streams[2])
    previous = None
    for f in range(frames):
        s = streams[2 + f % 2]
        if previous is not None:
            s.wait_event(previous)       # the application orders its own streams
        t.set_stream(s.cuda_stream)
        t.render(f)
        previous = torch.cuda.Event()
        previous.record(s)
    torch.cuda.synchronize()
    assert same(t.read_accum(), want)
    t.destroy()


def test_handles_driven_from_threads_of_their_own(pkg, oracle):
    """A handle is single-threaded, a process need not be: four host threads, each with a tracer and a stream of
    its own, render at the same time (one of the tracers gets the launch overlap) and every image is the oracle's."""
    import threading
    import torch
    w, h, frames = 200, 120, 16
    scene = pkg.default_scene()
    results, errors = {}, []

    def work(k):
        try:
            torch.cuda.set_device(0)
            stream = torch.cuda.Stream()
            t = make_tracer(pkg, w, h, 2 + k, 8, 0.05)
            t.set_stream(stream.cuda_stream)
            for f in range(frames):
                t.render(f)
            t.render_frames(frames, 5)
            results[k] = (t.read_accum(), t.get_params_words(0))
            t.destroy()
        except BaseException as e:      # reported by the main thread
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=100)
    assert not errors, errors
    assert len(results) == 4
    for k in range(4):
        ref = make_tracer(pkg, w, h, 2 + k, 8, 0.05)
        want = oracle_frames(oracle, ref, scene, range(frames + 5))["accum"]
        ref.destroy()
        assert same(results[k][0], want)


def test_scene_change_and_buffer_growth_behind_renders_in_flight(pkg, oracle):
    """rt_set_spheres and a growing per-frame-sum buffer wait for the handle's own kernels (an event on the
    caller's stream), not for the whole device: a scene written over the one a launch in flight still reads,
    or a freed buffer a fold still reads, would show in the image."""
    import torch
    w, h = 1920, 1080
    scene_a = pkg.default_scene()
    scene_b = scene_a.copy()
    scene_b[1:, 1] += 1.5                               # every small sphere lifted: most pixels change
    stream = torch.cuda.Stream()
    t = make_tracer(pkg, w, h, 2, 8, 0.05)
    t.set_stream(stream.cuda_stream)
    calm = make_tracer(pkg, w, h, 2, 8, 0.05)
    t.render_frames(0, 2)                            # small buffer of per-frame sums
    t.render_frames(2, 30)                           # grows it while the first call's fold may still run
    t.set_spheres(scene_b)                           # same count: written into the buffer the launch reads
    got = t.read_accum()
    calm.render_frames(0, 2)
    calm.synchronize()
    calm.render_frames(2, 30)
    calm.synchronize()
    assert same(got, calm.read_accum())
    band = oracle_frames(oracle, calm, scene_a, range(32), row_begin=536, row_end=540)
    assert same(got[536:540], band["accum"][536:540])
    t.render_frames(0, 2)                            # the raised flag clears; scene B from here on
    band = oracle_frames(oracle, t, scene_b, [0, 1], row_begin=536, row_end=544)
    assert same(t.read_accum()[536:544], band["accum"][536:544])
    t.destroy()
    calm.destroy()


@pytest.mark.parametrize("w,h", [(65536, 1), (1, 8192), (16384, 3), (7, 4099)])
def test_extreme_extents(pkg, oracle, w, h):
    """The widest image the interface accepts, one-pixel-wide columns, and sizes that are no multiple of any
    tile: every texel once, with the oracle's bits."""
    t = make_tracer(pkg, w, h, 1, 4, 0.05)
    t.enable_rng_capture(True)
    t.render_frames(0, 2)
    ref = oracle_frames(oracle, t, pkg.default_scene(), [0, 1], want_rng=True)
    assert same(t.read_rng_state(), ref["rng"])
    assert same(t.read_accum(), ref["accum"])
    assert same(t.read_output(), ref["output"])
    t.destroy()


@pytest.mark.parametrize("frame,spp", [(0xFFFFFFFF, 3), (0x55555555, 3), (0x7FFFFFFF, 2), (0xFFFFFFFE, 5)])
def test_frame_counter_wraps_like_the_shader(pkg, oracle, frame, spp):
    """(frame + 1) * spp is evaluated in 32-bit unsigned arithmetic by the shader (raytrace.comp.glsl:337): frame
    indices near 2^32 make the divisor wrap (to 0 for frame 0xFFFFFFFF: the image is then inf/NaN through the tone
    map), and the seed hash takes the frame index as it is."""
    w, h = 48, 20
    for overlap in (True, False):
        t = make_tracer(pkg, w, h, spp, 4, 0.05)
        t.set_frame_overlap(overlap)
        t.render(frame)
        ref = oracle_frames(oracle, t, pkg.default_scene(), [frame])
        assert same(t.read_accum(), ref["accum"])
        assert same(t.read_output(), ref["output"])
        # a batch that runs across the wrap: frames frame, frame + 1 (mod 2^32)
        t.reset_accumulation()
        t.render_frames(frame, 2)
        ref = oracle_frames(oracle, t, pkg.default_scene(), [frame, (frame + 1) & 0xFFFFFFFF])
        assert same(t.read_accum(), ref["accum"])
        assert same(t.read_output(), ref["output"])
        t.destroy()

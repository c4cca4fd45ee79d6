"""The benchmark configurations at their real parameters, HIP (through the C ABI) against the CPU oracle.

Full images at these sizes are out of the oracle's reach (3.4e10 samples for the 4K case), so each case
checks a few 8-row bands of the real image bit for bit -- pixel coordinates stay global, so a band of a
shard launch is the same computation as that band inside the full launch -- and the full-size launch
through properties that do not depend on the size: alpha, finiteness, and equality of the full launch's
bands with the banded launches that were checked against the oracle.  Needs a real MI355X."""
import numpy as np
import pytest

from test_gpu_parity import make_tracer, oracle_frames, same

pytestmark = pytest.mark.gpu


def _band_vs_oracle(pkg, oracle, t, scene, frames, bands):
    """Renders `frames` of each band through rt_render_shard and compares accumulator, RNG words and
    rgba8 of those rows with the oracle.  Returns {band: accumulator rows} for later comparisons."""
    t.enable_rng_capture(True)
    kept = {}
    for (y0, y1) in bands:
        t.reset_accumulation()
        t.render_shard(frames[0], len(frames), 1, y0, y1)
        acc, out, rng = t.read_accum()[y0:y1], t.read_output()[y0:y1], t.read_rng_state()[y0:y1]
        ref = oracle_frames(oracle, t, scene, frames, row_begin=y0, row_end=y1, want_rng=True)
        assert same(rng, ref["rng"][y0:y1])
        assert same(acc, ref["accum"][y0:y1])
        assert same(out, ref["output"][y0:y1])
        kept[(y0, y1)] = acc.copy()
        del ref
    return kept


def test_c1_720p_1spp_depth8(pkg, oracle):
    """The configuration quoted for a CPU run of the reference shader: default scene, 1280x720, 1 spp,
    depth 8, the application's aperture of 0.05, frame 0 -- the whole frame against the oracle: RNG words,
    accumulator, rgba8.  Then frames 1..3 on top of it (accumulation, one call per frame as the
    application drives it, overlapped launches) against the oracle's."""
    scene = pkg.default_scene()
    t = make_tracer(pkg, 1280, 720, 1, 8, 0.05)
    t.enable_rng_capture(True)
    t.render(0)
    ref = oracle_frames(oracle, t, scene, [0], want_rng=True)
    assert same(t.read_rng_state(), ref["rng"])
    assert same(t.read_accum(), ref["accum"])
    assert same(t.read_output(), ref["output"])
    t.enable_rng_capture(False)     # consecutive single-frame calls then overlap on the library's streams
    for f in (1, 2, 3):
        t.render(f)
    ref = oracle_frames(oracle, t, scene, [1, 2, 3], accum=ref["accum"])
    assert same(t.read_accum(), ref["accum"])
    assert same(t.read_output(), ref["output"])
    t.destroy()


def test_c3_4k_64spp_depth16_dof(pkg, oracle):
    """3840x2160, 64 spp per frame, depth 16, thin-lens depth of field on, accumulation to 4096 spp."""
    w, h = 3840, 2160
    scene = pkg.default_scene()
    t = make_tracer(pkg, w, h, 64, 16, 0.05)
    bands = [(0, 8), (1184, 1192), (2152, 2160)]      # sky, through the spheres, checkered ground
    two_frames = _band_vs_oracle(pkg, oracle, t, scene, [0, 1], bands)

    # the whole 64-frame accumulation in one call (8.5 GB of per-frame sums folded in frame order)
    t.reset_accumulation()
    t.render_frames(0, 64)
    full = t.read_accum()
    assert np.all(full[..., 3] == 1.0) and np.all(np.isfinite(full)) and np.all(full[..., :3] >= 0.0)
    full_out = t.read_output()
    full_rng = t.read_rng_state()
    # ... equals frame-by-frame dispatches (how the reference's caller drives render) on the checked bands
    s = make_tracer(pkg, w, h, 64, 16, 0.05)
    s.enable_rng_capture(True)
    for (y0, y1) in bands:
        for f in range(64):
            s.render_shard(f, 1, 1, y0, y1)
        acc = s.read_accum()[y0:y1]
        assert same(acc, full[y0:y1])
        assert same(s.read_output()[y0:y1], full_out[y0:y1])
        assert same(s.read_rng_state()[y0:y1], full_rng[y0:y1])
        if y0 == bands[1][0]:
            # and the first two of those single dispatches are the ones the oracle confirmed above
            s.render_shard(0, 1, 1, y0, y1)
            s.render_shard(1, 1, 1, y0, y1)
            assert same(s.read_accum()[y0:y1], two_frames[(y0, y1)])
    # converged image: the mean luminance of a band agrees with the oracle's 128-spp estimate of it
    y0, y1 = bands[1]
    mean_ref = two_frames[(y0, y1)][..., :3].mean() / 128.0
    mean_gpu = full[y0:y1, :, :3].mean() / 4096.0
    assert abs(mean_gpu - mean_ref) <= 0.02 * mean_ref
    s.destroy()
    t.destroy()


def test_c4_8k_256spp_row_sharded(pkg, oracle):
    """7680x4320, 256 spp, depth 8, one frame: bands against the oracle, then eight row shards and eight
    interleaved stripe shards (what eight GPUs would render) against the unsharded frame."""
    w, h = 7680, 4320
    scene = pkg.default_scene()
    t = make_tracer(pkg, w, h, 256, 8, 0.05)
    bands = [(2368, 2376), (4312, 4320)]
    checked = _band_vs_oracle(pkg, oracle, t, scene, [0], bands)

    t.reset_accumulation()
    t.render(0)
    whole, whole_out = t.read_accum(), t.read_output()
    assert np.all(whole[..., 3] == 1.0) and np.all(np.isfinite(whole))
    for (y0, y1), acc in checked.items():
        assert same(whole[y0:y1], acc)

    from raytracing_amd import dist as rtdist
    t.reset_accumulation()
    for (b, e) in rtdist.row_bands(h, 8):
        t.render_shard(0, 1, 1, b, e)
    assert same(t.read_accum(), whole)
    assert same(t.read_output(), whole_out)

    t.reset_accumulation()
    for rank in range(8):
        t.render_stripes(0, 1, 1, stripe_rows=16, stripe_first=rank, stripe_stride=8)
    assert same(t.read_accum(), whole)
    assert same(t.read_output(), whole_out)
    t.destroy()


def test_c5_4096_spheres_depth50_band(pkg, oracle):
    """The 4096-sphere scene at its real bounce budget (depth 50, 16 spp, 1920x1080): an 8-row band traced
    through the hierarchy against the oracle's loop over all spheres."""
    scene = pkg.synthetic_scene(4096)
    t = make_tracer(pkg, 1920, 1080, 16, 50, 0.05, scene=scene)
    t.set_camera([3.0, 1.5, 6.0], [-0.4, -0.2, -1.0], 6.0)
    checked = _band_vs_oracle(pkg, oracle, t, scene, [0], [(600, 608)])
    # the same band inside the full frame, and with the hierarchy switched off
    t.reset_accumulation()
    t.render(0)
    assert same(t.read_accum()[600:608], checked[(600, 608)])
    t.set_scene_hierarchy(False)
    t.reset_accumulation()
    t.render_shard(0, 1, 1, 600, 608)
    assert same(t.read_accum()[600:608], checked[(600, 608)])
    t.destroy()


def test_resolve_kernel_matches_oracle_tail(pkg, oracle):
    """rt_resolve_output alone (the display tail as its own kernel) on an accumulator written from the
    host: rgba8 must equal the oracle's tail bit for bit, including negative, huge, NaN and zero sums."""
    w, h = 256, 96
    rng = np.random.default_rng(21)
    acc = np.empty((h, w, 4), dtype=np.float32)
    acc[..., :3] = np.exp(rng.uniform(np.log(1e-4), np.log(1e5), size=(h, w, 3))).astype(np.float32)
    acc[..., 3] = 1.0
    acc[0, :8, 0] = np.float32([0.0, -0.0, -1.0, np.inf, np.nan, 1e-45, 3e38, 1.0])
    t = make_tracer(pkg, w, h, 4, 8)
    for sample_count in (1, 64, 4096, 0xFFFFFFFF):
        t.write_accum(acc)
        t.resolve_output(sample_count)
        assert same(t.read_output(), oracle.resolve(acc, sample_count))
    # a rendered accumulator: the tail on its own equals the tail fused into the dispatch
    t.reset_accumulation()
    t.render_frames(0, 3)
    fused = t.read_output()
    t.resolve_output(3 * 4)
    assert same(t.read_output(), fused)
    assert same(fused, oracle.resolve(t.read_accum(), 12))
    t.destroy()


@pytest.mark.parametrize("w,h,frames,stripe_rows,world", [(96, 70, 3, 8, 3), (64, 61, 1, 4, 4), (40, 33, 64, 6, 2),
                                                          (50, 20, 2, 16, 5), (33, 17, 5, 1, 8)])
def test_stripe_shards_equal_whole_image(pkg, oracle, w, h, frames, stripe_rows, world):
    """Interleaved stripes (rank r renders stripes r, r + G, ...) of every height, also heights that are no
    multiple of the kernel's pixel tile and a last stripe cut short by the image, put together are the
    unsharded image; the unsharded image is the oracle's."""
    scene = pkg.default_scene()
    t = make_tracer(pkg, w, h, 2, 6)
    t.enable_rng_capture(True)
    t.render_frames(0, frames)
    whole, whole_out, whole_rng = t.read_accum(), t.read_output(), t.read_rng_state()
    ref = oracle_frames(oracle, t, scene, range(frames), want_rng=True)
    assert same(whole, ref["accum"])
    s = make_tracer(pkg, w, h, 2, 6)
    s.enable_rng_capture(True)
    for rank in range(world):
        s.render_stripes(0, frames, 1, stripe_rows, rank, world)
    assert same(s.read_accum(), whole)
    assert same(s.read_output(), whole_out)
    assert same(s.read_rng_state(), whole_rng)
    # one rank's stripes only touch that rank's rows
    s.reset_accumulation()
    s.render_stripes(0, frames, 1, stripe_rows, 1 % world, world)
    got = s.read_accum()
    mine = np.zeros(h, dtype=bool)
    for k in range(1 % world, (h + stripe_rows - 1) // stripe_rows, world):
        mine[k * stripe_rows:(k + 1) * stripe_rows] = True
    assert same(got[mine], whole[mine])
    assert not got[~mine].any()
    t.destroy()
    s.destroy()


def test_failed_calls_leave_the_handle_usable(pkg, oracle):
    """A resize, a scene change or a batch that runs out of device memory (forced through the library's
    allocation-failure hook) reports an error and leaves extent, images and scene as they were."""
    lib = pkg.load_library()
    scene = pkg.default_scene()
    t = make_tracer(pkg, 64, 40, 2, 6)
    t.render(0)
    before = t.read_accum()
    try:
        for nth in (1, 2):                      # accumulator, then output image
            lib.rt_debug_fail_allocation(nth)
            with pytest.raises(RuntimeError, match="hipErrorOutOfMemory"):
                t.resize(128, 80)
            assert t.extent == (64, 40)
            assert same(t.read_accum(), before)
        big = pkg.synthetic_scene(300)          # needs a larger sphere buffer and a hierarchy
        for nth in (1, 2, 3):                   # sphere records, node geometry, node words
            lib.rt_debug_fail_allocation(nth)
            with pytest.raises(RuntimeError, match="hipErrorOutOfMemory"):
                t.set_spheres(big)
            assert same(t.get_spheres(), scene)
            assert t.get_scene_hierarchy()[0].shape[0] == 0
        lib.rt_debug_fail_allocation(1)         # the per-frame sums of a first multi-frame call
        with pytest.raises(RuntimeError, match="hipErrorOutOfMemory"):
            t.render_frames(1, 2)
    finally:
        lib.rt_debug_fail_allocation(0)
    # the handle still renders the old scene at the old extent, continuing the accumulation
    t.render(1)
    ref = oracle_frames(oracle, t, scene, [0, 1])
    assert same(t.read_accum(), ref["accum"])
    t.set_spheres(big)
    t.render_frames(0, 2)
    ref = oracle_frames(oracle, t, big, [0, 1])
    assert same(t.read_accum(), ref["accum"])
    t.destroy()

"""The small-scene kernel tests camera rays only against the spheres its unit's pixel tile can see.
This mirrors that decision in float32 and checks, for many rays of every tile, that a sphere left out
could not have been hit: hitSphere (shaders/raytrace.comp.glsl:109-133), evaluated in float32 like the
kernel, finds no root in [tMin, 1e20] for it."""
import numpy as np
import pytest

F = np.float32


def _dot(a, b):
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def _vec(block, n=3):
    return np.array(list(block)[:n], dtype=np.float32)


def _left_out(params, spheres, x0, y0, tw, th):
    """The kernel's estimate for one tile: True where a sphere is left out."""
    origin, lens_radius = _vec(params.origin_lens), F(params.origin_lens[3])
    ll, hz, vt = _vec(params.lower_left), _vec(params.horizontal), _vec(params.vertical)
    inv_w, inv_h, res_y = F(params.inv_resolution[0]), F(params.inv_resolution[1]), F(params.resolution[1])
    half_w, half_h = F(0.5) * F(tw), F(0.5) * F(th)
    uc = (F(x0) + half_w) * inv_w
    vc = ((res_y - F(y0)) - half_h) * inv_h
    dc = ll + hz * uc + vt * vc - origin
    lens = F(1.5) * abs(lens_radius)
    rho = ((half_w * inv_w) * np.sqrt(_dot(hz, hz)) + (half_h * inv_h) * np.sqrt(_dot(vt, vt)) + lens) * F(1.001)
    w = spheres[:, 0:3] - origin
    r2 = spheres[:, 3] * spheres[:, 3]
    cr = np.cross(w, dc).astype(np.float32)
    len_dc, len_w = np.sqrt(_dot(dc, dc)), np.sqrt(_dot(w, w))
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        q_c = np.sqrt(_dot(cr, cr)) / len_dc
        reach = (np.sqrt(r2) + lens) * F(1.001)
        narrow = rho < F(0.25) * len_dc
        misses = narrow & (q_c - len_w * (F(1.6) * rho / len_dc) - F(0.002) * len_w > reach)
        behind = narrow & (len_w - lens > F(1.01) * np.sqrt(r2)) & \
            (_dot(w, dc) + len_w * rho + (F(0.002) * len_w + lens) * len_dc < F(0.0))
    return misses | behind


def _tile_rays(params, x0, y0, tw, th, rng, count):
    """Camera rays of the tile as :317-328 build them (float32), jitter and lens sample at random and at
    their extremes."""
    origin, lens_radius = _vec(params.origin_lens), F(params.origin_lens[3])
    ll, hz, vt = _vec(params.lower_left), _vec(params.horizontal), _vec(params.vertical)
    cam_u, cam_v = _vec(params.u), _vec(params.v)
    inv_w, inv_h, res_y = F(params.inv_resolution[0]), F(params.inv_resolution[1]), F(params.resolution[1])
    px = rng.integers(x0, x0 + tw, count).astype(np.float32)
    py = rng.integers(y0, y0 + th, count).astype(np.float32)
    top = F(1.0) - F(2.0 ** -24)
    jx = rng.choice([F(0.0), top, F(0.5)], count) * (rng.random(count) < 0.5) + rng.random(count, dtype=np.float32) * (rng.random(count) < 0.5)
    jy = rng.choice([F(0.0), top, F(0.5)], count) * (rng.random(count) < 0.5) + rng.random(count, dtype=np.float32) * (rng.random(count) < 0.5)
    jx = np.minimum(jx.astype(np.float32), top)
    jy = np.minimum(jy.astype(np.float32), top)
    u = ((px + jx) * inv_w)[:, None]
    v = ((((res_y - F(1.0)) - py) + jy) * inv_h)[:, None]
    ang = rng.random(count) * 2.0 * np.pi
    rad = np.sqrt(rng.random(count)) * (rng.random(count) < 0.7) + (rng.random(count) < 0.3) * 0.999999
    disk = np.stack([np.cos(ang) * rad, np.sin(ang) * rad], 1).astype(np.float32)
    rdx = (lens_radius * disk[:, 0])[:, None]
    rdy = (lens_radius * disk[:, 1])[:, None]
    offset = cam_u * rdx + cam_v * rdy
    d = ll + hz * u + vt * v - origin - offset
    o = origin + offset
    with np.errstate(invalid="ignore", divide="ignore"):
        d = d * (F(1.0) / np.sqrt(_dot(d, d)))[:, None]          # normalize = v * inversesqrt(dot(v, v))
    return o.astype(np.float32), d.astype(np.float32)


def _roots_accepted(o, d, spheres):
    oc = o[:, None, :] - spheres[None, :, 0:3]
    a = _dot(d, d)[:, None]
    hb = _dot(oc, d[:, None, :])
    c = _dot(oc, oc) - (spheres[:, 3] * spheres[:, 3])[None, :]
    disc = hb * hb - a * c
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        sq = np.sqrt(np.where(disc < 0, F(0.0), disc))
        near = (-hb - sq) / a
        far = (-hb + sq) / a
    near_out = (near < F(0.001)) | (near > F(1e20))
    far_out = (far < F(0.001)) | (far > F(1e20))
    return ~(disc < 0) & ~(near_out & far_out)


CASES = [
    # (scene, camera position, direction, focus, aperture, fov, width, height)
    ("default", [13.0, 2.0, 3.0], [-13.0, -1.0, -3.0], 13.4, 0.0, 20.0, 96, 54),
    ("default", [13.0, 2.0, 3.0], [-13.0, -1.0, -3.0], 13.4, 0.3, 20.0, 96, 54),
    ("default", [0.5, 1.0, 2.5], [-0.1, 0.0, -1.0], 2.0, 0.05, 90.0, 64, 40),     # close to the centre sphere
    ("default", [0.0, 1.0, 0.2], [1.0, 0.2, 0.0], 1.0, 0.0, 120.0, 40, 24),       # inside the centre sphere
    ("default", [30.0, 0.02, 0.0], [-1.0, 0.0, 0.0], 30.0, 0.0, 30.0, 128, 32),   # skimming the ground
    ("default", [0.0, 400.0, 0.0], [0.001, -1.0, 0.0], 400.0, 2.0, 5.0, 48, 48),  # far above, looking down
    ("random16", [6.0, 3.0, 6.0], [-1.0, -0.4, -1.0], 8.0, 0.1, 60.0, 80, 48),
    ("random16", [-2.0, 0.5, 9.0], [0.3, 0.1, -1.0], 8.0, 0.0, 100.0, 33, 17),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_left_out_spheres_cannot_be_hit(pkg, case):
    name, pos, look, focus, aperture, fov, width, height = CASES[case]
    rng = np.random.default_rng(100 + case)
    if name == "default":
        scene = pkg.default_scene()
    else:
        scene = np.zeros((16, 12), dtype=np.float32)
        scene[:, 0:3] = rng.normal(size=(16, 3)) * 3.0
        scene[:, 3] = 10.0 ** rng.uniform(-1.5, 0.5, 16) * rng.choice([1.0, -1.0], 16, p=[0.8, 0.2])
        scene[0] = [0.0, -1000.0, 0.0, 1000.0, 0.5, 0.5, 0.5, 0.0, 0.0, 0.0, 1.0, 1.0]
    t = pkg.RayTracer().create(width, height, device_id=-1)
    t.set_aperture(aperture)
    t.set_fov(fov)
    t.set_camera(pos, look, focus)
    params = t.get_params(0)
    t.destroy()
    left_out_total = 0
    tests = 0
    for th_log, tw_log in ((3, 3), (2, 1)):                      # single-frame and many-frame tiles
        tw, th = 1 << tw_log, 1 << th_log
        for y0 in range(0, height, th):
            for x0 in range(0, width, tw):
                w_, h_ = min(tw, width - x0), min(th, height - y0)
                out = _left_out(params, scene, x0, y0, w_, h_)
                if not out.any():
                    continue
                o, d = _tile_rays(params, x0, y0, w_, h_, rng, 96)
                hit = _roots_accepted(o, d, scene)
                assert not np.any(hit[:, out]), (x0, y0, np.nonzero(out)[0])
                left_out_total += int(out.sum())
                tests += 1
    if case in (0, 4, 6):
        assert left_out_total > tests       # the estimate does leave spheres out


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(len(CASES)))
def test_cameras_match_the_oracle(pkg, oracle, case):
    """The same views rendered by the HIP path, one frame at a time (8 x 8 pixel tiles) and as a batch of
    frames (2 x 4 pixel tiles), against the oracle."""
    from test_gpu_parity import oracle_frames, same
    name, pos, look, focus, aperture, fov, width, height = CASES[case]
    rng = np.random.default_rng(100 + case)
    if name == "default":
        scene = pkg.default_scene()
    else:
        scene = np.zeros((16, 12), dtype=np.float32)
        scene[:, 0:3] = rng.normal(size=(16, 3)) * 3.0
        scene[:, 3] = 10.0 ** rng.uniform(-1.5, 0.5, 16) * rng.choice([1.0, -1.0], 16, p=[0.8, 0.2])
        scene[:, 4:7] = rng.random((16, 3))
        scene[:, 8] = rng.integers(0, 3, 16)
        scene[:, 9] = 0.2
        scene[:, 10] = 1.5
        scene[0] = [0.0, -1000.0, 0.0, 1000.0, 0.5, 0.5, 0.5, 0.0, 0.0, 0.0, 1.0, 1.0]
    for frames in (1, 9):
        t = pkg.RayTracer().create(width, height, device_id=0)
        t.set_samples_per_pixel(3)
        t.set_max_depth(6)
        t.set_aperture(aperture)
        t.set_fov(fov)
        t.set_spheres(scene)
        t.set_camera(pos, look, focus)
        t.enable_rng_capture(True)
        t.render_frames(0, frames)
        ref = oracle_frames(oracle, t, scene, range(frames), want_rng=True)
        assert same(t.read_rng_state(), ref["rng"])
        assert same(t.read_accum(), ref["accum"])
        t.destroy()

"""Randomised parity of the HIP path with the CPU oracle: scenes, cameras and settings drawn from a
seeded generator, every case compared bit for bit (RNG words, accumulator floats, rgba8 bytes).
The cases are small so that the oracle finishes in a fraction of a second each."""
import numpy as np
import pytest

from test_gpu_parity import oracle_frames, same

pytestmark = pytest.mark.gpu


def _random_scene(rng, count):
    """Spheres of mixed sizes and materials scattered around the origin; radii from 0.01 to 30,
    overlapping and nested spheres included, a big ground sphere in half of the scenes; material
    codes beyond 2 (treated as dielectric by the shader's else branch) and negative radii appear too."""
    s = np.zeros((count, 12), dtype=np.float32)
    s[:, 0:3] = rng.normal(size=(count, 3)) * rng.choice([0.5, 3.0, 12.0])
    s[:, 3] = 10.0 ** rng.uniform(-2.0, 1.0, size=count) * rng.choice([1.0, 1.0, 1.0, -1.0], size=count)
    s[:, 4:7] = rng.random((count, 3))
    s[:, 8] = rng.choice([0.0, 0.0, 1.0, 2.0, 3.0], size=count)
    s[:, 9] = rng.random(count) * rng.choice([0.0, 0.3, 1.0])
    s[:, 10] = rng.choice([1.0, 1.3, 1.5, 2.4, 0.7], size=count)
    s[:, 11] = rng.choice([0.0, 1.0], size=count)
    if rng.random() < 0.5:
        s[0] = [0.0, -1000.0, 0.0, 1000.0, 0.6, 0.6, 0.6, 0.0, 0.0, 0.0, 1.0, 1.0]
    if count > 3 and rng.random() < 0.5:
        s[2] = s[1]                      # coincident pair: exact ties
        s[2, 8] = 1.0
    return s


SPECIALS = [0.0, -0.0, np.nan, np.inf, -np.inf, 1e30, -1e30, 1e-30, 1e-42, 3.0e38, 2.0 ** 20, 2.0 ** 21, 2.0 ** -20, 2.0 ** -21]


BIG_COUNTS = [401, 1024, 1025, 1500, 2500, 4096, 4097, 5600, 6000, 9000]


def _case(seed, specials=False, big=False):
    """big: scenes of 401 to 9000 spheres on small images -- whole records in LDS up to 1024 spheres, geometry
    only up to 4096, global memory beyond, the hierarchy in LDS up to about 5600 spheres and in global memory
    beyond, each with the hierarchy on or off."""
    rng = np.random.default_rng(seed)
    count = int(rng.choice(BIG_COUNTS if big else [1, 2, 5, 17, 23, 24, 25, 63, 64, 65, 150, 400]))
    scene = _random_scene(rng, count)
    if specials:
        # out-of-range and non-finite values in geometry and material fields: the shader's arithmetic is
        # defined for them too (NaN compares false, so e.g. a NaN root is accepted by :125-131)
        for _ in range(int(rng.integers(1, 4))):
            scene[int(rng.integers(0, count)), int(rng.choice([0, 1, 2, 3, 4, 9, 10]))] = rng.choice(SPECIALS)
    if big:
        return {
            "scene": scene,
            "w": int(rng.integers(3, 28)), "h": int(rng.integers(2, 18)),
            "spp": int(rng.choice([1, 2])), "depth": int(rng.choice([1, 3, 12, 50])),
            "aperture": float(rng.choice([0.0, 0.05, 0.8])), "fov": float(rng.uniform(5.0, 120.0)),
            "frames": int(rng.choice([1, 2, 3])),
            "pos": (rng.normal(size=3) * rng.choice([2.0, 15.0, 80.0])).astype(np.float32),
            "focus": float(10.0 ** rng.uniform(-1.0, 2.0)),
            "pattern": str(rng.choice(["batch", "calls", "stripes"])),
            "ranks": int(rng.integers(1, 4)), "stripe_rows": int(rng.choice([1, 2, 4, 8])),
            # 0: the loop over all spheres; 1: the hierarchy as the library walks it by default (the four-way tree
            # where it fits LDS); 2: the depth-first tree, one node per step
            "hierarchy": (1 + seed % 2) if rng.random() < 0.6 else 0,
        }
    return {
        "scene": scene,
        "w": int(rng.integers(3, 70)), "h": int(rng.integers(2, 50)),
        "spp": int(rng.choice([1, 2, 3, 5])), "depth": int(rng.choice([1, 2, 5, 12, 50])),
        "aperture": float(rng.choice([0.0, 0.05, 0.8])), "fov": float(rng.uniform(5.0, 120.0)),
        "frames": int(rng.choice([1, 2, 3, 7, 40])),
        "pos": (rng.normal(size=3) * rng.choice([2.0, 15.0, 80.0])).astype(np.float32),
        "focus": float(10.0 ** rng.uniform(-1.0, 2.0)),
        # how the frames are submitted: one launch, one call per frame (consecutive launches overlap on the
        # library's two streams), or as the interleaved stripe shards of several ranks
        "pattern": str(rng.choice(["batch", "batch", "calls", "stripes"])),
        "ranks": int(rng.integers(1, 6)), "stripe_rows": int(rng.choice([1, 2, 3, 4, 8, 16])),
        "hierarchy": 2 if seed % 3 == 0 else 1,
    }


@pytest.mark.parametrize("seed", range(64))
def test_random_case_matches_the_oracle(pkg, oracle, seed):
    _run_case(pkg, oracle, _case(1000 + seed))


@pytest.mark.parametrize("seed", range(32))
def test_special_values_match_the_oracle(pkg, oracle, seed):
    _run_case(pkg, oracle, _case(5000 + seed, specials=True))


@pytest.mark.parametrize("seed", range(12))
def test_large_random_scenes_match_the_oracle(pkg, oracle, seed):
    _run_case(pkg, oracle, _case(9000 + seed, specials=(seed % 4 == 3), big=True))


def _run_case(pkg, oracle, c):
    t = pkg.RayTracer().create(c["w"], c["h"], device_id=0)
    t.set_samples_per_pixel(c["spp"])
    t.set_max_depth(c["depth"])
    t.set_aperture(c["aperture"])
    t.set_fov(c["fov"])
    t.set_spheres(c["scene"])
    t.set_scene_hierarchy(int(c.get("hierarchy", 1)))
    target = c["scene"][len(c["scene"]) // 2, 0:3]
    direction = target - c["pos"]
    if not np.any(direction != 0.0):
        direction = np.array([0.0, 0.0, -1.0], dtype=np.float32)
    t.set_camera(c["pos"].tolist(), direction.tolist(), c["focus"])
    ref = oracle_frames(oracle, t, c["scene"], range(c["frames"]), want_rng=True)
    if c["pattern"] == "batch":
        t.enable_rng_capture(True)
        t.render_frames(0, c["frames"])
        assert same(t.read_rng_state(), ref["rng"])
    elif c["pattern"] == "calls":
        for f in range(c["frames"]):
            t.render(f)
    else:
        for rank in range(c["ranks"]):
            t.render_stripes(0, c["frames"], 1, c["stripe_rows"], rank, c["ranks"])
    assert same(t.read_accum(), ref["accum"])
    assert same(t.read_output(), ref["output"])
    t.destroy()

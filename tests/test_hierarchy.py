"""The bounding-sphere hierarchy for large scenes: structure, the guarantee that it never skips a
sphere the plain loop (shaders/raytrace.comp.glsl:247-259) would have accepted, and GPU parity with
the loop over all spheres."""
import numpy as np
import pytest

F = np.float32
LEAF = 0x80000000
KEEP = F(1.0) - F(2.0 ** -13)


def _host(pkg, scene):
    t = pkg.RayTracer().create(64, 64, device_id=-1)
    t.set_spheres(scene)
    return t


def _cluster_scene(pkg, count=300, seed=5):
    """Tight clumps, nested and coincident spheres, and a wide range of radii."""
    rng = np.random.default_rng(seed)
    out = np.zeros((count, 12), dtype=np.float32)
    out[:, 4:7] = rng.random((count, 3), dtype=np.float32)
    out[:, 8] = rng.integers(0, 3, count).astype(np.float32)
    out[:, 9] = rng.random(count, dtype=np.float32) * 0.3
    out[:, 10] = 1.5
    centres = rng.normal(size=(12, 3)).astype(np.float32) * 6.0
    for i in range(count):
        c = centres[i % 12]
        out[i, 0:3] = c + rng.normal(size=3).astype(np.float32) * np.float32(0.8)
        out[i, 3] = np.float32(10.0 ** rng.uniform(-2.0, 0.3))
    out[0] = [0.0, -500.0, 0.0, 495.0, 0.5, 0.5, 0.5, 0.0, 0.0, 0.0, 1.0, 1.0]
    out[10:20] = out[20:30]                      # coincident spheres: exact ties, the later index must win
    out[10:20, 4:7] = 0.9
    out[40:50, 0:3] = out[50:60, 0:3]            # concentric, different radii
    return out


def _check_structure(scene, geom, word):
    n = len(word)
    leaf = (word & LEAF) != 0
    assert leaf.sum() == len(scene)
    index = (word[leaf] & 0x7FFFFFFF) >> 17
    assert sorted(index.tolist()) == list(range(len(scene)))
    # a leaf holds the sphere's centre and the float product r * r, and continues at the next node
    s = scene[index]
    assert np.array_equal(geom[leaf, :3], s[:, :3])
    assert np.array_equal(geom[leaf, 3], s[:, 3] * s[:, 3])
    at = np.nonzero(leaf)[0]
    assert np.array_equal((word[leaf] & 0x1FFFF), ((at + 1) * 16).astype(np.uint32))
    # an inner node's ball (R^2 = w / 1.125) contains every sphere of its subtree, which ends where
    # its offset points
    for i in np.nonzero(~leaf)[0]:
        assert word[i] % 16 == 0
        end = int(word[i]) // 16
        assert i + 1 < end <= n
        sub = np.arange(i + 1, end)
        sl = sub[leaf[sub]]
        assert len(sl) >= 2
        centre = geom[i, :3].astype(np.float64)
        radius = np.sqrt(float(geom[i, 3]) / 1.125)
        c = geom[sl, :3].astype(np.float64)
        r = np.sqrt(geom[sl, 3].astype(np.float64))
        assert np.all(np.sqrt(((c - centre) ** 2).sum(1)) + r <= radius * (1.0 + 1e-6))
        # subtrees nest: a child's subtree ends inside its parent's
        inner = sub[~leaf[sub]]
        assert np.all(word[inner] // 16 <= end)


@pytest.mark.parametrize("name", ["synthetic4096", "synthetic100", "random", "clusters"])
def test_structure(pkg, name):
    scene = {"synthetic4096": lambda: pkg.synthetic_scene(4096), "synthetic100": lambda: pkg.synthetic_scene(100),
             "random": pkg.random_scene, "clusters": lambda: _cluster_scene(pkg)}[name]()
    t = _host(pkg, scene)
    geom, word = t.get_scene_hierarchy()
    assert 0 < len(word) <= 7850
    _check_structure(scene, geom, word)
    # the same scene gives the same hierarchy
    t2 = _host(pkg, scene)
    geom2, word2 = t2.get_scene_hierarchy()
    assert np.array_equal(geom, geom2) and np.array_equal(word, word2)
    t.destroy()
    t2.destroy()


def test_scenes_without_hierarchy(pkg):
    t = pkg.RayTracer().create(64, 64, device_id=-1)
    assert len(t.get_scene_hierarchy()[1]) == 0           # the default five spheres
    t.set_spheres(pkg.synthetic_scene(23))
    assert len(t.get_scene_hierarchy()[1]) == 0           # below the threshold
    t.set_spheres(pkg.synthetic_scene(24))
    assert len(t.get_scene_hierarchy()[1]) > 0
    for field, value in ((0, np.nan), (1, np.inf), (2, 3.0e6), (3, 0.0), (3, 1.0e-25), (3, 2.0e20), (3, np.nan)):
        scene = pkg.synthetic_scene(200)
        scene[77, field] = value
        t.set_spheres(scene)
        assert len(t.get_scene_hierarchy()[1]) == 0, (field, value)
    t.set_spheres(pkg.synthetic_scene(200))
    assert len(t.get_scene_hierarchy()[1]) > 0
    t.destroy()


def _dot(a, b):
    # left-to-right float32 sum of products, like the kernel's dot()
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def _quadratic(centre, w, o, d, a, scale):
    oc = o - centre
    hb = _dot(oc, d)
    c = _dot(oc, oc) * scale - w
    disc = hb * hb - a * c
    return hb, c, disc


def _passes(hb, c, disc):
    """The two tests ahead of the square root: negative discriminant; centre behind the origin of a
    ray that starts outside (neither half_b nor c has its sign bit set)."""
    behind = ((hb.view(np.uint32) | c.view(np.uint32)) & np.uint32(0x80000000)) == 0
    return ~((disc < F(0.0)) | behind)


def _accepted(hb, disc, a, tmin=F(0.001), tmax=F(1e20)):
    """hitSphere's root selection (raytrace.comp.glsl:121-131) in float32: whether the sphere yields a
    root inside [tmin, tmax].  1e20 is the loop's initial `closest`, so this is every sphere the loop
    over all spheres could ever accept."""
    with np.errstate(invalid="ignore"):
        sq = np.sqrt(np.where(disc < F(0.0), F(0.0), disc))
        near = (-hb - sq) / a
        far = (-hb + sq) / a
    near_out = (near < tmin) | (near > tmax)
    far_out = (far < tmin) | (far > tmax)
    return ~(disc < F(0.0)) & ~(near_out & far_out)


def _visited_leaves(geom, word, o, d):
    """Walks the hierarchy for every ray at once, in float32 like the kernel; returns a (rays, nodes)
    mask of the leaves each ray reached."""
    n = len(word)
    rays = o.shape[0]
    a = _dot(d, d)
    at = np.zeros(rays, dtype=np.int64)
    visited = np.zeros((rays, n), dtype=bool)
    leaf = (word & LEAF) != 0
    skip = (word & 0x1FFFF).astype(np.int64) // 16
    lanes = np.arange(rays)
    for _ in range(n + 1):
        live = at < n
        if not live.any():
            break
        idx = np.where(live, at, 0)
        scale = np.where(leaf[idx], F(1.0), KEEP).astype(np.float32)
        hb, c, disc = _quadratic(geom[idx, :3], geom[idx, 3], o, d, a, scale)
        ok = _passes(hb, c, disc)
        visited[lanes[live & leaf[idx]], idx[live & leaf[idx]]] = True
        at = np.where(live, np.where(ok, at + 1, skip[idx]), at)
    else:
        raise AssertionError("a traversal did not terminate")
    return visited


def _rays(scene, count, seed):
    """Camera-like, surface-born, far-away and grazing rays; directions of assorted lengths."""
    rng = np.random.default_rng(seed)
    o = np.zeros((count, 3), dtype=np.float32)
    d = np.zeros((count, 3), dtype=np.float32)
    for i in range(count):
        kind = i % 6
        s = scene[rng.integers(0, len(scene))]
        target = s[:3] + rng.normal(size=3) * s[3]
        if kind == 0:      # from a camera position
            origin = np.array([3.0, 1.5, 6.0]) + rng.normal(size=3) * 0.01
        elif kind == 1:    # from the surface of another sphere
            q = scene[rng.integers(0, len(scene))]
            n = rng.normal(size=3)
            n /= np.linalg.norm(n)
            origin = q[:3] + n * q[3] * 1.001
        elif kind == 2:    # from far away
            n = rng.normal(size=3)
            origin = n / np.linalg.norm(n) * 10.0 ** rng.uniform(1.5, 4.0)
        elif kind == 3:    # grazing: aimed at the silhouette of a sphere
            origin = np.array([3.0, 1.5, 6.0]) + rng.normal(size=3) * 3.0
            to = s[:3] - origin
            side = np.cross(to, rng.normal(size=3))
            side /= np.linalg.norm(side)
            target = s[:3] + side * s[3] * (1.0 + rng.normal() * 1e-6)
        elif kind == 4:    # from inside a cluster, any direction
            origin = s[:3] + rng.normal(size=3) * 2.0
            target = origin + rng.normal(size=3)
        else:              # parallel to one or two axes (zero components of either sign), along a face of a sphere's
            # bounding box or through its centre: the slab test divides by these components
            axes = rng.permutation(3)[:int(rng.integers(1, 3))]
            offset = np.zeros(3)
            for k in range(3):
                if k not in axes:
                    offset[k] = s[3] * rng.choice([-1.0, 0.0, 1.0, rng.normal()])
            origin = s[:3].astype(np.float64) + offset
            direction = np.zeros(3)
            for k in axes:
                direction[k] = rng.choice([-1.0, 1.0]) * 10.0 ** rng.uniform(-2.0, 2.0)
                origin[k] -= direction[k] / abs(direction[k]) * (s[3] + 10.0 ** rng.uniform(-1.0, 1.5))
            o[i] = origin
            d[i] = direction * rng.choice([1.0, 1.0, -1.0])          # some point away from the sphere
            d[i][d[i] == 0.0] = rng.choice([0.0, -0.0], size=int((d[i] == 0.0).sum()))
            continue
        direction = target - origin
        direction *= 10.0 ** rng.uniform(-2.0, 2.0) / np.linalg.norm(direction)
        o[i] = origin
        d[i] = direction
    return o, d


@pytest.mark.parametrize("name", ["synthetic4096", "random", "clusters"])
def test_no_accepted_sphere_is_skipped(pkg, name):
    """For every ray, each sphere whose float32 evaluation of hitSphere yields a root in [tmin, 1e20]
    is a leaf the walk reaches (and is then evaluated exactly as in the loop over all spheres)."""
    scene = {"synthetic4096": lambda: pkg.synthetic_scene(4096), "random": pkg.random_scene,
             "clusters": lambda: _cluster_scene(pkg)}[name]()
    t = _host(pkg, scene)
    geom, word = t.get_scene_hierarchy()
    t.destroy()
    leaf_nodes = np.nonzero((word & LEAF) != 0)[0]
    sphere_of = ((word[leaf_nodes] & 0x7FFFFFFF) >> 17).astype(np.int64)
    node_of_sphere = np.zeros(len(scene), dtype=np.int64)
    node_of_sphere[sphere_of] = leaf_nodes
    rays = 1500 if name == "synthetic4096" else 3000
    o, d = _rays(scene, rays, seed=11)
    a = _dot(d, d)
    assert np.all((a >= F(2.0 ** -20)) & (a <= F(2.0 ** 20)))
    visited = _visited_leaves(geom, word, o, d)
    centre = scene[:, :3]
    r2 = scene[:, 3] * scene[:, 3]
    missed = 0
    reached = 0
    candidates = 0
    for lo in range(0, rays, 250):
        sl = slice(lo, lo + 250)
        hb, c, disc = _quadratic(centre[None, :, :], r2[None, :], o[sl, None, :], d[sl, None, :], a[sl, None], F(1.0))
        ok = _accepted(hb, disc, a[sl, None])
        seen = visited[sl][:, node_of_sphere]
        missed += int((ok & ~seen).sum())
        candidates += int(ok.sum())
        reached += int(seen.sum())
    assert missed == 0
    assert candidates > rays // 2                  # the rays do meet spheres
    assert reached < 0.2 * rays * len(scene)       # and the walk does prune


def _roots(hb, disc, a, tmin, tmax):
    """The root hitSphere accepts inside [tmin, tmax] (near first, then far) and whether there is one."""
    with np.errstate(invalid="ignore", over="ignore"):
        sq = np.sqrt(np.where(disc < F(0.0), F(0.0), disc))
        near = (-hb - sq) / a
        far = (-hb + sq) / a
    near_out = (near < tmin) | (near > tmax)
    far_out = (far < tmin) | (far > tmax)
    root = np.where(near_out, far, near)
    return root, ~(disc < F(0.0)) & ~(near_out & far_out)


def _walk_closest(geom, word, o, d):
    """The kernel's walk in float32 for every ray at once, with its running `closest` and the test that
    leaves out balls lying wholly beyond it; returns (closest, index of the winning sphere or -1)."""
    n = len(word)
    rays = o.shape[0]
    a = _dot(d, d)
    at = np.zeros(rays, dtype=np.int64)
    leaf = (word & LEAF) != 0
    skip = (word & 0x1FFFF).astype(np.int64) // 16
    sphere = ((word & 0x7FFFFFFF) >> 17).astype(np.int64)
    closest = np.full(rays, 1e20, dtype=np.float32)
    best = np.full(rays, -1, dtype=np.int64)
    limit = np.full(rays, np.inf, dtype=np.float32)
    visits = 0
    for _ in range(n + 1):
        live = at < n
        if not live.any():
            break
        visits += int(live.sum())
        idx = np.where(live, at, 0)
        scale = np.where(leaf[idx], F(1.0), KEEP).astype(np.float32)
        hb, c, disc = _quadratic(geom[idx, :3], geom[idx, 3], o, d, a, scale)
        with np.errstate(invalid="ignore", over="ignore"):
            ok = _passes(hb, c, disc) & ~(c - geom[idx, 3] > limit)
            root, found = _roots(hb, disc, a, F(0.001), closest)
            take = live & ok & leaf[idx] & found & ((root < closest) | (sphere[idx] > best))
            closest = np.where(take, root, closest)
            best = np.where(take, sphere[idx], best)
            limit = np.where(take, F(2.5) * a * closest * closest, limit)
        at = np.where(live, np.where(ok, at + 1, skip[idx]), at)
    else:
        raise AssertionError("a traversal did not terminate")
    return closest, best, visits


@pytest.mark.parametrize("name", ["synthetic4096", "random", "clusters"])
def test_walk_finds_the_hit_of_the_loop_over_all_spheres(pkg, name):
    """The walk, with the balls beyond the running closest hit left out, ends with the same root and the
    same sphere as the loop of :247-259 (smallest accepted root; of equal roots the highest index)."""
    scene = {"synthetic4096": lambda: pkg.synthetic_scene(4096), "random": pkg.random_scene,
             "clusters": lambda: _cluster_scene(pkg)}[name]()
    t = _host(pkg, scene)
    geom, word = t.get_scene_hierarchy()
    t.destroy()
    rays = 1500 if name == "synthetic4096" else 3000
    o, d = _rays(scene, rays, seed=23)
    a = _dot(d, d)
    closest, best, visits = _walk_closest(geom, word, o, d)
    centre = scene[:, :3]
    r2 = scene[:, 3] * scene[:, 3]
    hits = 0
    for lo in range(0, rays, 250):
        sl = slice(lo, lo + 250)
        hb, c, disc = _quadratic(centre[None, :, :], r2[None, :], o[sl, None, :], d[sl, None, :], a[sl, None], F(1.0))
        root, found = _roots(hb, disc, a[sl, None], F(0.001), F(1e20))
        root = np.where(found, root, F(np.inf))
        want_closest = root.min(axis=1)
        any_hit = np.isfinite(want_closest)
        ties = root == want_closest[:, None]
        want_best = np.where(any_hit, len(scene) - 1 - np.argmax(ties[:, ::-1], axis=1), -1)
        want_closest = np.where(any_hit, want_closest, F(1e20)).astype(np.float32)
        assert np.array_equal(best[sl], want_best)
        assert np.array_equal(closest[sl].view(np.uint32), want_closest.view(np.uint32))
        hits += int(any_hit.sum())
    assert hits > rays // 2
    assert visits < 0.25 * rays * len(word)    # node visits, inner nodes included


# ------------------------------------------------------------------------ the four-way tree
NONE = 0xFFFFFFFF


def _quad(pkg, scene):
    t = _host(pkg, scene)
    q = t.get_scene_quadtree()
    t.destroy()
    return q


UNUSED_BOX = F(3.0e38)


def _check_quad_structure(scene, q):
    box, geom, slot_sphere = q["box"], q["geom"], q["slot_sphere"]
    groups, levels, roots, first_leaf = q["groups"], q["levels"], q["roots"], q["first_leaf"]
    assert levels >= 1 and 1 <= roots <= 16
    per_level = [roots * 4 ** l for l in range(levels)]
    assert groups == sum(per_level) + 1                          # the tree, then the prologue
    assert first_leaf == sum(per_level[:-1]) == len(box)
    assert len(geom) == groups - first_leaf
    assert len(slot_sphere) == 4 * (groups - first_leaf)
    assert levels <= 9                                              # four bits of pending siblings per level below the roots
    used = slot_sphere != NONE
    assert sorted(slot_sphere[used].tolist()) == list(range(len(scene)))     # every sphere in exactly one slot
    leaf_geom = geom.reshape(-1, 4)
    s = scene[slot_sphere[used]]
    assert np.array_equal(leaf_geom[used, :3], s[:, :3])
    assert np.array_equal(leaf_geom[used, 3], s[:, 3] * s[:, 3])             # the float product the loop uses
    assert np.all(np.isneginf(leaf_geom[~used, 3]))
    assert q["prologue_spheres"] == int(used[-4:].sum()) and np.all(used[-4:][:q["prologue_spheres"]])
    # spheres are dealt out evenly: the groups of the last level differ by at most one sphere
    per_group = used[:-4].reshape(-1, 4).sum(1)
    assert per_group.max() - per_group.min() <= 1 and per_group.max() >= 1

    def spheres_below(g, level):
        if level == levels - 1:
            return [int(x) for x in slot_sphere[4 * (g - first_leaf):4 * (g - first_leaf) + 4] if x != NONE]
        out = []
        for k in range(4):
            below = spheres_below(4 * g + roots + k, level + 1)
            lo, hi = box[g, 0:3, k].astype(np.float64), box[g, 3:6, k].astype(np.float64)
            if not below:
                assert np.all(box[g, :, k] == UNUSED_BOX)
                continue
            c = scene[below, :3].astype(np.float64)
            r = np.sqrt((scene[below, 3] * scene[below, 3]).astype(np.float64))          # the radius the kernel's r^2 stands for
            assert np.all(c - r[:, None] >= lo) and np.all(c + r[:, None] <= hi)         # contains every sphere below
            assert np.all((c - r[:, None]).min(0) - lo <= 1e-6 * (1.0 + np.abs(lo)))     # and is tight
            assert np.all(hi - (c + r[:, None]).max(0) <= 1e-6 * (1.0 + np.abs(hi)))
            out += below
        return out

    in_tree = []
    for root in range(roots):
        in_tree += spheres_below(root, 0)
    assert sorted(in_tree + [int(x) for x in slot_sphere[-4:] if x != NONE]) == list(range(len(scene)))
    # the ball and the smallest radius the walk's margin is derived from cover the spheres of the tree
    c = scene[in_tree, :3].astype(np.float64)
    r = np.sqrt((scene[in_tree, 3] * scene[in_tree, 3]).astype(np.float64))
    assert np.all(np.sqrt(((c - q["bound_centre"].astype(np.float64)) ** 2).sum(1)) + r <= q["bound_radius"])
    assert 0.0 < q["min_radius"] <= r.min()


@pytest.mark.parametrize("name", ["synthetic4096", "synthetic100", "synthetic24", "synthetic4097", "synthetic20000", "random",
                                  "clusters"])
def test_quadtree_structure(pkg, name):
    scene = {"synthetic4096": lambda: pkg.synthetic_scene(4096), "synthetic100": lambda: pkg.synthetic_scene(100),
             "synthetic24": lambda: pkg.synthetic_scene(24), "synthetic4097": lambda: pkg.synthetic_scene(4097),
             "synthetic20000": lambda: pkg.synthetic_scene(20000),
             "random": pkg.random_scene, "clusters": lambda: _cluster_scene(pkg)}[name]()
    q = _quad(pkg, scene)
    assert q["groups"] > 0
    _check_quad_structure(scene, q)
    q2 = _quad(pkg, scene)
    assert all(np.array_equal(q[k].view(np.uint32), q2[k].view(np.uint32)) for k in ("box", "geom", "slot_sphere"))
    if name == "synthetic4096":
        # the ground sphere is kept out of the tree; 4095 spheres fill 1024 groups of the sixth level
        assert q["prologue_spheres"] == 1 and q["slot_sphere"][-4] == 0 and q["levels"] == 6 and q["roots"] == 1


def test_scenes_without_quadtree(pkg):
    t = pkg.RayTracer().create(64, 64, device_id=-1)
    assert t.get_scene_quadtree()["groups"] == 0           # the default five spheres
    t.set_spheres(pkg.synthetic_scene(23))
    assert t.get_scene_quadtree()["groups"] == 0
    scene = pkg.synthetic_scene(200)
    scene[77, 0] = np.nan
    t.set_spheres(scene)
    assert t.get_scene_quadtree()["groups"] == 0
    t.set_spheres(pkg.synthetic_scene(9000))                  # does not fit LDS: the same kind of tree, walked in global memory
    q = t.get_scene_quadtree()
    assert q["groups"] > 0 and 96 * q["first_leaf"] + 64 * len(q["geom"]) > 152 * 1024
    assert len(t.get_scene_hierarchy_links()[1]) > 0
    t.destroy()


def _fma(a, b, c):
    """fma in binary32 for the box test (the product of two binary32 values is exact in binary64; the second
    rounding of the sum can differ from a true fma in the last bit, far inside the margin the test is about)."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def _slab_constants(q, o, d):
    """Per ray: 1/d clamped to +-2^90 and the two offsets -(o +- m) / d of the kernel's box test, m the ray's margin."""
    with np.errstate(divide="ignore", over="ignore"):
        inv = np.clip(F(1.0) / d, F(-2.0 ** 90), F(2.0 ** 90)).astype(np.float32)
    fc = o - q["bound_centre"][None, :]
    reach = np.sqrt(_dot(fc, fc)) * F(1.0001) + F(q["bound_radius"])
    k = F(26.0) * F(2.0 ** -24) / F(q["min_radius"])
    margin = (reach * np.minimum(F(0.003), reach * k + F(2e-6)) + F(3e-7) * np.abs(o).max(axis=-1)).astype(np.float32)
    lo_off = (-(o + margin[:, None]) * inv).astype(np.float32)
    hi_off = (-(o - margin[:, None]) * inv).astype(np.float32)
    return inv, lo_off, hi_off, margin


def _box_interval(lo, hi, inv, lo_off, hi_off):
    """[tn, tf] of the kernel's slab test before the clamps to 0 and `closest`; lo/hi (..., 3)."""
    a = _fma(lo, inv, lo_off)
    b = _fma(hi, inv, hi_off)
    # The kernel takes no minimum or maximum per axis: it reads the bound the ray crosses first (the high one where the
    # sign bit of inv is set) and the one it crosses last, and the two parameters are ordered by construction.
    neg = np.signbit(np.broadcast_to(inv, a.shape))
    near, far = np.where(neg, b, a), np.where(neg, a, b)
    assert np.all(near <= far), "a slab's two parameters came out in the wrong order"
    return near.max(axis=-1), far.min(axis=-1)


def _quad_walk_closest(q, scene, o, d):
    """The kernel's walk of the four-way tree in float32, ray by ray: prologue spheres first, four boxes per visited
    group, down to the child the ray enters first, pending siblings as four bits per level, spheres of a last-level
    group resolved in slot order with ties going to the higher sphere index."""
    box, geom, slot_sphere = q["box"], q["geom"], q["slot_sphere"]
    roots, first_leaf, groups = q["roots"], q["first_leaf"], q["groups"]
    rays = o.shape[0]
    out_closest = np.full(rays, 1e20, dtype=np.float32)
    out_best = np.full(rays, -1, dtype=np.int64)
    inv_all, lo_off_all, hi_off_all, _ = _slab_constants(q, o, d)
    visits = 0
    for ray in range(rays):
        ro, rd = o[ray], d[ray]
        a = _dot(rd, rd)
        inv, lo_off, hi_off = inv_all[ray], lo_off_all[ray], hi_off_all[ray]
        state = {"closest": F(1e20), "best": -1}

        def spheres(rel_group, only=4):
            balls = geom[rel_group][:only]
            with np.errstate(invalid="ignore", over="ignore"):
                hb, c, disc = _quadratic(balls[:, :3], balls[:, 3], ro[None, :], rd[None, :], a, F(1.0))
                ok = _passes(hb, c, disc)
            for k in np.nonzero(ok)[0]:
                with np.errstate(invalid="ignore", over="ignore"):
                    root_t, found = _roots(hb[k:k + 1], disc[k:k + 1], a, F(0.001), state["closest"])
                if not found[0]:
                    continue
                slot = 4 * rel_group + int(k)
                if root_t[0] < state["closest"] or state["best"] < 0 or slot_sphere[slot] > slot_sphere[state["best"]]:
                    state["closest"], state["best"] = root_t[0], slot

        spheres(groups - 1 - first_leaf, q["prologue_spheres"])
        stack, root, grp = 0, 0, 0
        while grp is not None:
            visits += 1
            done = False
            if grp < first_leaf:
                lo, hi = box[grp, 0:3, :].T, box[grp, 3:6, :].T          # (child, axis)
                with np.errstate(invalid="ignore", over="ignore"):
                    tn, tf = _box_interval(lo, hi, inv[None, :], lo_off[None, :], hi_off[None, :])
                enter = np.maximum(tn, F(0.0))
                ok = enter <= np.minimum(tf, state["closest"])
                mask = int(sum(1 << k for k in range(4) if ok[k]))
                if mask:
                    # first the child the ray enters first (the kernel's key: the bits of the entry parameter with
                    # the child's number in the two lowest), the others in the order of their numbers
                    keys = [((int(enter[k].view(np.uint32)) & ~3) | k) if ok[k] else 0xFFFFFFFF for k in range(4)]
                    k = min(keys) & 3
                    stack = (stack << 4) | (mask & ~(1 << k))
                    assert stack < 2 ** 32
                    grp = 4 * grp + roots + k
                else:
                    done = True
            if not done and grp >= first_leaf:
                spheres(grp - first_leaf)
                done = True
            if done:
                if stack == 0:
                    root += 1
                    grp = root if root < roots else None
                else:
                    up = ((stack & -stack).bit_length() - 1) >> 2
                    stack >>= 4 * up
                    for _ in range(up):
                        grp = (grp - roots) >> 2
                    k = (stack & -stack).bit_length() - 1
                    stack &= stack - 1
                    grp = ((grp - roots) & ~3) + roots + k
        out_closest[ray] = state["closest"]
        out_best[ray] = -1 if state["best"] < 0 else int(slot_sphere[state["best"]])
    return out_closest, out_best, visits


@pytest.mark.parametrize("name", ["synthetic4096", "synthetic100", "synthetic20000", "random", "clusters"])
def test_quad_walk_finds_the_hit_of_the_loop_over_all_spheres(pkg, name):
    """The walk of the four-way tree ends with the same root and the same sphere as the loop of :247-259
    (smallest accepted root; of equal roots the highest index), for camera-like, surface-born, far-away and
    grazing rays of assorted lengths."""
    scene = {"synthetic4096": lambda: pkg.synthetic_scene(4096), "synthetic100": lambda: pkg.synthetic_scene(100),
             "synthetic20000": lambda: pkg.synthetic_scene(20000),
             "random": pkg.random_scene, "clusters": lambda: _cluster_scene(pkg)}[name]()
    q = _quad(pkg, scene)
    rays = 400 if len(scene) > 1000 else 800
    o, d = _rays(scene, rays, seed=29)
    a = _dot(d, d)
    closest, best, visits = _quad_walk_closest(q, scene, o, d)
    centre = scene[:, :3]
    r2 = scene[:, 3] * scene[:, 3]
    hits = 0
    step = 50 if len(scene) > 10000 else 200
    for lo in range(0, rays, step):
        sl = slice(lo, lo + step)
        hb, c, disc = _quadratic(centre[None, :, :], r2[None, :], o[sl, None, :], d[sl, None, :], a[sl, None], F(1.0))
        root, found = _roots(hb, disc, a[sl, None], F(0.001), F(1e20))
        root = np.where(found, root, F(np.inf))
        want_closest = root.min(axis=1)
        any_hit = np.isfinite(want_closest)
        ties = root == want_closest[:, None]
        want_best = np.where(any_hit, len(scene) - 1 - np.argmax(ties[:, ::-1], axis=1), -1)
        want_closest = np.where(any_hit, want_closest, F(1e20)).astype(np.float32)
        assert np.array_equal(best[sl], want_best)
        assert np.array_equal(closest[sl].view(np.uint32), want_closest.view(np.uint32))
        hits += int(any_hit.sum())
    assert hits > rays // 2
    assert visits < (0.04 if len(scene) > 1000 else 0.6) * rays * q["groups"]       # group visits: the walk does prune


@pytest.mark.parametrize("name", ["synthetic4096", "random", "clusters"])
def test_accepted_roots_lie_inside_the_grown_boxes(pkg, name):
    """What the box walk rests on: every root that the float32 evaluation of hitSphere accepts for a sphere lies
    inside the parameter interval [tn, tf] that the kernel's slab test computes for ANY box containing the sphere
    (checked against the sphere's own bounding box, the smallest such box), with room to spare: the interval of a
    box grown by only a third of the ray's margin already contains it."""
    scene = {"synthetic4096": lambda: pkg.synthetic_scene(4096), "random": pkg.random_scene,
             "clusters": lambda: _cluster_scene(pkg)}[name]()
    q = _quad(pkg, scene)
    in_tree = np.sort(q["slot_sphere"][:-4][q["slot_sphere"][:-4] != NONE]).astype(np.int64)
    rays = 1500 if name == "synthetic4096" else 3000
    o, d = _rays(scene, rays, seed=53)
    a = _dot(d, d)
    inv, lo_off, hi_off, margin = _slab_constants(q, o, d)
    with np.errstate(divide="ignore", over="ignore"):
        third = (margin * F(1.0 / 3.0))[:, None]
        lo_off3 = (-(o + third) * inv).astype(np.float32)
        hi_off3 = (-(o - third) * inv).astype(np.float32)
    centre = scene[in_tree, :3]
    radius = scene[in_tree, 3]
    r2 = radius * radius
    r_seen = np.sqrt(r2.astype(np.float64))[:, None]                    # the radius the kernel's r^2 stands for
    lo_box = centre.astype(np.float64) - r_seen
    hi_box = centre.astype(np.float64) + r_seen
    lo_box = np.nextafter(lo_box.astype(np.float32), F(-np.inf))       # rounded outward, as the builder does
    hi_box = np.nextafter(hi_box.astype(np.float32), F(np.inf))
    accepted = 0
    for first in range(0, rays, 100):
        sl = slice(first, first + 100)
        hb, c, disc = _quadratic(centre[None, :, :], r2[None, :], o[sl, None, :], d[sl, None, :], a[sl, None], F(1.0))
        for tmax in (F(1e20), None):
            if tmax is None:
                # with a closest hit in force: the smallest accepted root of the ray
                root, found = _roots(hb, disc, a[sl, None], F(0.001), F(1e20))
                tmax_rays = np.where(found, root, F(np.inf)).min(axis=1)
                tmax_rays = np.where(np.isfinite(tmax_rays), tmax_rays, F(1e20)).astype(np.float32)[:, None]
            else:
                tmax_rays = np.full((hb.shape[0], 1), tmax, dtype=np.float32)
            root, found = _roots(hb, disc, a[sl, None], F(0.001), tmax_rays)
            found &= _passes(hb, c, disc)
            with np.errstate(invalid="ignore", over="ignore"):
                tn, tf = _box_interval(lo_box[None, :, :], hi_box[None, :, :], inv[sl, None, :], lo_off3[sl, None, :],
                                       hi_off3[sl, None, :])
            inside = (tn <= root) & (root <= tf)
            assert np.all(inside[found])
            # and the test as the kernel applies it, with the full margin, keeps the box
            with np.errstate(invalid="ignore", over="ignore"):
                tn, tf = _box_interval(lo_box[None, :, :], hi_box[None, :, :], inv[sl, None, :], lo_off[sl, None, :],
                                       hi_off[sl, None, :])
            reached = np.maximum(tn, F(0.0)) <= np.minimum(tf, tmax_rays)
            assert np.all(reached[found])
            accepted += int(found.sum())
    assert accepted > rays


# ---------------------------------------------------------------------------------------- GPU
def _render(pkg, scene, hierarchy, w=160, h=96, spp=4, depth=12, frames=3, camera=None, focus=6.0, aperture=0.05):
    t = pkg.RayTracer().create(w, h, device_id=0)
    t.set_samples_per_pixel(spp)
    t.set_max_depth(depth)
    t.set_aperture(aperture)
    t.set_spheres(scene)
    pos, look = camera or ([3.0, 1.5, 6.0], [-0.4, -0.2, -1.0])
    t.set_camera(pos, look, focus)
    t.set_scene_hierarchy(hierarchy)
    t.enable_rng_capture(True)
    t.render_frames(0, frames)
    out = (t.read_accum(), t.read_rng_state(), t.read_output())
    t.destroy()
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["synthetic4096", "synthetic700", "random", "clusters"])
def test_hierarchy_matches_loop_on_gpu(pkg, name):
    scene = {"synthetic4096": lambda: pkg.synthetic_scene(4096), "synthetic700": lambda: pkg.synthetic_scene(700),
             "random": pkg.random_scene, "clusters": lambda: _cluster_scene(pkg)}[name]()
    camera = ([13.0, 2.0, 3.0], [-13.0, -1.0, -3.0]) if name == "random" else None
    if name == "clusters":
        camera = ([0.0, 4.0, 30.0], [0.0, -0.1, -1.0])
    a = _render(pkg, scene, True, camera=camera)      # the four-way tree (all four scenes fit LDS)
    b = _render(pkg, scene, False, camera=camera)     # the loop over all spheres
    c = _render(pkg, scene, 2, camera=camera)         # the depth-first tree, one node per step
    for x, y, z in zip(a, b, c):
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8))
        assert np.array_equal(z.view(np.uint8), y.view(np.uint8))
    assert np.any(a[0][..., :3] > 0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("walk", [1, 2])
def test_hierarchy_ties_and_oracle(pkg, oracle, walk):
    """Coincident spheres give exactly equal roots; the later index must win as in the shader's loop."""
    from test_gpu_parity import oracle_frames, same
    scene = _cluster_scene(pkg, count=120)
    t = pkg.RayTracer().create(96, 64, device_id=0)
    t.set_samples_per_pixel(2)
    t.set_max_depth(8)
    t.set_spheres(scene)
    t.set_scene_hierarchy(walk)
    t.set_camera([0.0, 4.0, 30.0], [0.0, -0.1, -1.0], 30.0)
    t.enable_rng_capture(True)
    assert len(t.get_scene_hierarchy()[1]) > 0
    t.render_frames(0, 2)
    ref = oracle_frames(oracle, t, scene, [0, 1], want_rng=True)
    assert same(t.read_rng_state(), ref["rng"])
    assert same(t.read_accum(), ref["accum"])
    t.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("walk", [1, 2])
def test_rays_outside_the_proven_range_use_the_loop(pkg, walk):
    """A focus distance of 4000 makes the camera rays' dot(d,d) exceed 2^20: those rays must take the
    plain loop inside the hierarchy kernel, and the image must not change."""
    scene = pkg.synthetic_scene(500)
    scene[100:110, :4] = scene[110:120, :4]       # coincident spheres: ties inside that loop as well
    a = _render(pkg, scene, walk, focus=4000.0, aperture=0.0)
    b = _render(pkg, scene, False, focus=4000.0, aperture=0.0)
    for x, y in zip(a, b):
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,frames,spp", [(67, 35, 70, 1), (33, 9, 5, 3), (130, 3, 1, 2), (5, 5, 129, 1)])
def test_hierarchy_unit_boundaries(pkg, w, h, frames, spp):
    """Ragged tiles, more frames than one chunk holds, single frames: the waves of the hierarchy kernel
    move from unit to unit while some of their lanes are still busy, and nothing may change."""
    scene = pkg.synthetic_scene(300)
    a = _render(pkg, scene, True, w=w, h=h, spp=spp, depth=6, frames=frames)
    b = _render(pkg, scene, False, w=w, h=h, spp=spp, depth=6, frames=frames)
    c = _render(pkg, scene, 2, w=w, h=h, spp=spp, depth=6, frames=frames)
    for x, y, z in zip(a, b, c):
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8))
        assert np.array_equal(z.view(np.uint8), y.view(np.uint8))


@pytest.mark.gpu
def test_hierarchy_frame_by_frame_and_row_shards(pkg):
    """One launch per frame (as the application renders), and the same frames rendered as row shards,
    against one batched launch."""
    scene = pkg.synthetic_scene(300)

    def tracer():
        t = pkg.RayTracer().create(96, 50, device_id=0)
        t.set_samples_per_pixel(2)
        t.set_max_depth(8)
        t.set_spheres(scene)
        t.set_camera([3.0, 1.5, 6.0], [-0.4, -0.2, -1.0], 6.0)
        return t

    batch = tracer()
    batch.render_frames(0, 6)
    single = tracer()
    for f in range(6):
        single.render(f)
    shards = tracer()
    shards.render_shard(0, 6, 1, 0, 17, write_output=False)
    shards.render_shard(0, 6, 1, 17, 50, write_output=False)
    want = batch.read_accum()
    assert np.array_equal(single.read_accum().view(np.uint32), want.view(np.uint32))
    assert np.array_equal(shards.read_accum().view(np.uint32), want.view(np.uint32))
    for t in (batch, single, shards):
        t.destroy()


def _check_links(scene, geom, link, sphere):
    n = len(link)
    leaf = sphere != 0xFFFFFFFF
    assert leaf.sum() == len(scene)
    assert sorted(sphere[leaf].tolist()) == list(range(len(scene)))
    s = scene[sphere[leaf]]
    assert np.array_equal(geom[leaf, :3], s[:, :3])
    assert np.array_equal(geom[leaf, 3], s[:, 3] * s[:, 3])
    at = np.nonzero(leaf)[0]
    assert np.array_equal(link[leaf], (at + 1).astype(np.uint32))
    inner = np.nonzero(~leaf)[0]
    assert np.all(link[inner] > inner + 1) and np.all(link[inner] <= n)
    # every node's ball (R^2 = w / 1.125) holds the spheres of its subtree; checked through the chain
    # child ball + distance <= parent ball would be too strict (balls are built from the spheres), so
    # sample inner nodes instead
    rng = np.random.default_rng(1)
    for i in rng.choice(inner, size=min(len(inner), 300), replace=False):
        sub = np.arange(i + 1, link[i])
        sl = sub[leaf[sub]]
        centre = geom[i, :3].astype(np.float64)
        radius = np.sqrt(float(geom[i, 3]) / 1.125)
        c = geom[sl, :3].astype(np.float64)
        r = np.sqrt(geom[sl, 3].astype(np.float64))
        assert np.all(np.sqrt(((c - centre) ** 2).sum(1)) + r <= radius * (1.0 + 1e-6))
        assert np.all(link[sub[~leaf[sub]]] <= link[i])


def test_links_agree_with_packed_words_and_cover_large_scenes(pkg):
    small = pkg.synthetic_scene(500)
    t = _host(pkg, small)
    geom, word = t.get_scene_hierarchy()
    geom2, link, sphere = t.get_scene_hierarchy_links()
    assert np.array_equal(geom, geom2)
    assert np.array_equal(word & 0x1FFFF, link * 16)
    leaf = (word & LEAF) != 0
    assert np.array_equal(leaf, sphere != 0xFFFFFFFF)
    assert np.array_equal(((word[leaf] & 0x7FFFFFFF) >> 17), sphere[leaf])
    _check_links(small, geom2, link, sphere)
    big = pkg.synthetic_scene(9000)
    t.set_spheres(big)
    geom, link, sphere = t.get_scene_hierarchy_links()
    assert len(link) > 7850                      # beyond the packed format
    _check_links(big, geom, link, sphere)
    with pytest.raises(RuntimeError, match="too large for packed"):
        t.get_scene_hierarchy()
    t.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("count", [6000, 20000])
def test_scenes_beyond_lds(pkg, oracle, count):
    """More spheres than LDS holds: the hierarchy is walked in global memory, the loop over all spheres
    reads its geometry records from global memory; both against the oracle."""
    from test_gpu_parity import oracle_frames, same
    scene = pkg.synthetic_scene(count)
    results = []
    for hierarchy in (True, False):
        t = pkg.RayTracer().create(40, 24, device_id=0)
        t.set_samples_per_pixel(2)
        t.set_max_depth(6)
        t.set_spheres(scene)
        t.set_scene_hierarchy(hierarchy)
        t.set_camera([3.0, 1.5, 6.0], [-0.4, -0.2, -1.0], 6.0)
        t.enable_rng_capture(True)
        t.render_frames(0, 3)
        results.append((t.read_accum(), t.read_rng_state(), t.read_output()))
        if hierarchy:
            ref = oracle_frames(oracle, t, scene, range(3), want_rng=True)
        t.destroy()
    for x, y in zip(*results):
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8))
    assert same(results[0][1], ref["rng"])
    assert same(results[0][0], ref["accum"])
    assert same(results[0][2], ref["output"])


@pytest.mark.gpu
def test_4096_spheres_at_full_size(pkg):
    """The 4096-sphere configuration at its full size (1920x1080, 16 spp, depth 50, one frame): the
    hierarchy kernel (persistent waves, resumable walks) against the loop over all spheres."""
    scene = pkg.synthetic_scene(4096)
    a = _render(pkg, scene, True, w=1920, h=1080, spp=16, depth=50, frames=1)
    b = _render(pkg, scene, False, w=1920, h=1080, spp=16, depth=50, frames=1)
    c = _render(pkg, scene, 2, w=1920, h=1080, spp=16, depth=50, frames=1)
    for x, y, z in zip(a, b, c):
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8))
        assert np.array_equal(z.view(np.uint8), y.view(np.uint8))


@pytest.mark.gpu
def test_largest_tree_kept_in_lds(pkg):
    """A scene whose hierarchy is just below the node limit of the packed format: its nodes and the hash
    tables of sixteen waves must fit the LDS of a compute unit."""
    count = 5600
    scene = pkg.synthetic_scene(count)
    t = _host(pkg, scene)
    nodes = len(t.get_scene_hierarchy_links()[1])
    while nodes > 7850:
        count -= 50
        scene = pkg.synthetic_scene(count)
        t.set_spheres(scene)
        nodes = len(t.get_scene_hierarchy_links()[1])
    assert nodes > 7600 and len(t.get_scene_hierarchy()[1]) == nodes      # packed
    t.destroy()
    a = _render(pkg, scene, 2, w=64, h=40, spp=1, depth=4, frames=2)
    b = _render(pkg, scene, False, w=64, h=40, spp=1, depth=4, frames=2)
    for x, y in zip(a, b):
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8))


@pytest.mark.gpu
def test_largest_four_way_tree_kept_in_lds(pkg):
    """The largest scene whose four-way tree still fits LDS (152 KiB beside the hash tables of sixteen waves), and the
    first one beyond, whose tree is walked in global memory."""
    def tree_bytes(q):
        return 96 * q["first_leaf"] + 64 * len(q["geom"])

    count = 7300
    t = _host(pkg, pkg.synthetic_scene(count))
    while tree_bytes(t.get_scene_quadtree()) > 152 * 1024:
        count -= 100
        t.set_spheres(pkg.synthetic_scene(count))
    assert 130 * 1024 < tree_bytes(t.get_scene_quadtree())
    t.destroy()
    for n in (count, count + 100):
        scene = pkg.synthetic_scene(n)
        a = _render(pkg, scene, True, w=64, h=40, spp=1, depth=4, frames=2)
        b = _render(pkg, scene, False, w=64, h=40, spp=1, depth=4, frames=2)
        for x, y in zip(a, b):
            assert np.array_equal(x.view(np.uint8), y.view(np.uint8))


def test_ancestor_of_a_group_in_closed_form():
    """The walk goes up `up` levels at once: the parent of group g is (g - roots) >> 2, and the kernel takes the
    ancestor `up` levels higher as (g - roots * (4^up - 1) / 3) >> (2 up), with (4^up - 1) / 3 read off 0x55555555.
    Checked against the step-by-step parent for every group of trees of every shape the builder makes."""
    for roots in range(1, 17):
        for levels in range(1, 7):
            first = [0]
            for level in range(levels):
                first.append(first[-1] + roots * 4 ** level)          # first group of each level, and the group count
            groups = np.arange(first[-1], dtype=np.int64)
            level_of = np.searchsorted(np.asarray(first[1:]), groups, side="right")
            for up in range(0, levels):
                sel = groups[level_of >= up]
                step = sel.copy()
                for _ in range(up):
                    step = (step - roots) >> 2
                e = 0x55555555 & ((1 << (2 * up)) - 1)
                assert e == (4 ** up - 1) // 3
                closed = (sel - roots * e) >> (2 * up)
                assert np.array_equal(step, closed)
                assert np.array_equal(np.searchsorted(np.asarray(first[1:]), closed, side="right"), level_of[level_of >= up] - up)

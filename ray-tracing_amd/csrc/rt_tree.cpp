// Host-side construction of the bounding-sphere hierarchy described in rt_tree.hpp.
//
// Why the inflation makes skipping safe (u = 2^-24, the unit roundoff of binary32; all claims are
// for rays whose dot(d,d) lies in [2^-20, 2^20] and whose origin is finite and below 2^30 in
// magnitude -- the kernel sends every other ray through the plain loop):
//
//  * hitSphere computes disc = hb*hb - a*(dot(oc,oc) - r^2) with about ten roundings; its value
//    differs from the exact a*(r^2 - q^2), q = distance of the centre from the ray's line, by
//    less than 32u * a * |oc|^2.  A sphere can therefore only pass the discriminant test when
//    q^2 < r^2 + 32u |oc|^2.
//  * An internal node (centre C, all spheres below it inside the ball of radius R) is tested
//    with the same expression, with w = 1.125 R^2 in place of r^2 and dot(oc,oc) scaled by
//    k = 1 - 2^-13.  If that test fails in floating point, the exact distance Q of C from the line
//    satisfies Q^2 >= 1.12 R^2 + 2020u |oc|^2, so s = Q - R obeys s^2 >= 216u |oc|^2
//    (from (R+s)^2 <= 1.12 R^2 + 9.34 s^2).  Every sphere below the node then has q - r >= s and
//    |oc_i| <= 2 |oc|, hence q^2 - r^2 >= s^2 >= 216u |oc|^2 > 32u |oc_i|^2: its discriminant is
//    negative as computed, and the plain loop would have rejected it too.
//  * A node whose ball lies behind the ray origin (origin outside the inflated ball, hb >= 0) is
//    skipped as well: every sphere inside the ball meets the line only at parameters below
//    -0.05 |oc| / |d|, far further from zero than the rounding of the roots (below 0.003 |oc| / |d|),
//    so both roots come out negative and :125-131 reject them.
//  * A ball (centre at distance D from the origin, radius R; a leaf's own sphere or a node's ball)
//    is also left out when it lies wholly beyond the closest hit found so far, L = closest * |d|
//    away.  The walk tests (k D^2 - w) - w > 2.5 a closest^2 with the values it has anyway, i.e.
//    D^2 - 2.25 R^2 > 2.5 L^2 for a node and D^2 - 2 r^2 > 2.5 L^2 for a leaf, up to a relative 2^-13.
//    Since 2 r^2 + 2.5 L^2 >= 1.11 (L + r)^2, that gives D > 1.05 (L + R): every point of the ball is
//    more than 1.05 L + 0.05 R from the origin, while the computed roots of a sphere inside it are
//    off by less than 0.003 (D + R) / |d| even for a grazing hit.  So both roots exceed `closest`
//    and :125-131 would reject them (an equal root, which replaces an earlier sphere, cannot occur).
#include "rt_tree.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace rt {

namespace {

struct Item {
    double c[3];
    double r;       // radius the kernel's arithmetic sees: sqrt of the float product r*r, nudged up
    uint32_t index;
};

struct Builder {
    std::vector<Item> items;
    SphereTree* out;

    uint32_t emit(const float g[4], uint32_t link, uint32_t sphere)
    {
        const uint32_t at = out->node_count();
        out->geom.insert(out->geom.end(), g, g + 4);
        out->link.push_back(link);
        out->sphere.push_back(sphere);
        return at;
    }

    void emit_leaf(const Item& it, const rt_gpu_sphere* spheres)
    {
        const float* cr = spheres[it.index].center_radius;
        const float g[4] = { cr[0], cr[1], cr[2], cr[3] * cr[3] }; // the kernel's staging computes the same product
        emit(g, out->node_count() + 1u, it.index);
    }

    // Ball around the midpoint of the bounding box of items[lo, hi); centre rounded to float first.
    void bounding_ball(uint32_t lo, uint32_t hi, float g[4]) const
    {
        double mn[3] = { 1e300, 1e300, 1e300 }, mx[3] = { -1e300, -1e300, -1e300 };
        for (uint32_t i = lo; i < hi; ++i) {
            for (int k = 0; k < 3; ++k) {
                mn[k] = std::min(mn[k], items[i].c[k] - items[i].r);
                mx[k] = std::max(mx[k], items[i].c[k] + items[i].r);
            }
        }
        double centre[3];
        for (int k = 0; k < 3; ++k) {
            g[k] = static_cast<float>(0.5 * (mn[k] + mx[k]));
            centre[k] = g[k];
        }
        double radius = 0.0;
        for (uint32_t i = lo; i < hi; ++i) {
            double d2 = 0.0;
            for (int k = 0; k < 3; ++k) {
                const double d = items[i].c[k] - centre[k];
                d2 += d * d;
            }
            radius = std::max(radius, std::sqrt(d2) + items[i].r);
        }
        radius *= 1.0 + 1e-9; // the double arithmetic above
        g[3] = std::nextafterf(static_cast<float>(1.125 * radius * radius), INFINITY);
    }

    // Splits items[lo, hi) in two along the longest axis of the centres' bounding box, at the
    // spatial midpoint, or at the median when the midpoint leaves one side empty.
    uint32_t split(uint32_t lo, uint32_t hi)
    {
        double mn[3] = { 1e300, 1e300, 1e300 }, mx[3] = { -1e300, -1e300, -1e300 };
        for (uint32_t i = lo; i < hi; ++i) {
            for (int k = 0; k < 3; ++k) {
                mn[k] = std::min(mn[k], items[i].c[k]);
                mx[k] = std::max(mx[k], items[i].c[k]);
            }
        }
        int axis = 0;
        for (int k = 1; k < 3; ++k) {
            if (mx[k] - mn[k] > mx[axis] - mn[axis]) {
                axis = k;
            }
        }
        const double mid = 0.5 * (mn[axis] + mx[axis]);
        auto first = items.begin() + lo, last = items.begin() + hi;
        auto cut = std::partition(first, last, [&](const Item& it) { return it.c[axis] < mid; });
        uint32_t at = static_cast<uint32_t>(cut - items.begin());
        if (at == lo || at == hi) {
            at = lo + (hi - lo) / 2;
            std::nth_element(first, items.begin() + at, last, [&](const Item& a, const Item& b) {
                return a.c[axis] != b.c[axis] ? a.c[axis] < b.c[axis] : a.index < b.index;
            });
        }
        return at;
    }

    void emit_part(uint32_t lo, uint32_t hi, const rt_gpu_sphere* spheres)
    {
        if (hi - lo == 1) {
            emit_leaf(items[lo], spheres);
            return;
        }
        float g[4];
        bounding_ball(lo, hi, g);
        const uint32_t at = emit(g, 0u, kTreeNoSphere);
        emit_children(lo, hi, spheres);
        out->link[at] = out->node_count();
    }

    // Children of a node (or the top level): up to four parts, each a leaf or a node of its own;
    // four or fewer spheres become leaves directly.
    void emit_children(uint32_t lo, uint32_t hi, const rt_gpu_sphere* spheres)
    {
        if (hi - lo <= 4) {
            for (uint32_t i = lo; i < hi; ++i) {
                emit_leaf(items[i], spheres);
            }
            return;
        }
        const uint32_t m = split(lo, hi);
        uint32_t cuts[5] = { lo, m, m, hi, hi };
        int parts = 2;
        if (hi - lo > 8) {
            cuts[1] = split(lo, m);
            cuts[2] = m;
            cuts[3] = split(m, hi);
            cuts[4] = hi;
            parts = 4;
        } else {
            cuts[2] = hi;
        }
        for (int p = 0; p < parts; ++p) {
            if (cuts[p + 1] > cuts[p]) {
                emit_part(cuts[p], cuts[p + 1], spheres);
            }
        }
    }
};


// The spheres as the builders see them; false when the scene does not qualify for a hierarchy.
bool collect_items(const rt_gpu_sphere* spheres, uint32_t count, std::vector<Item>& items)
{
    if (spheres == nullptr || count < kTreeMinSpheres || count > kTreeMaxSpheres) {
        return false;
    }
    items.resize(count);
    for (uint32_t i = 0; i < count; ++i) {
        const float* cr = spheres[i].center_radius;
        const float r2 = cr[3] * cr[3];
        for (int k = 0; k < 3; ++k) {
            if (!(std::fabs(cr[k]) <= 0x1p20f)) {
                return false; // also rejects NaN
            }
            items[i].c[k] = cr[k];
        }
        if (!(r2 >= 0x1p-40f && r2 <= 0x1p40f)) {
            return false;
        }
        items[i].r = std::sqrt(static_cast<double>(r2)) * (1.0 + 1e-9);
        items[i].index = i;
    }
    return true;
}

// Builder of the complete four-way tree (rt_tree.hpp).
//
// Why the walk's box test never leaves out a sphere that the loop over all spheres would have accepted (same
// ranges as above; u = 2^-24):
//
//  * Let t be a root that hitSphere computes and accepts for sphere i (centre c, radius r, oc = o - c,
//    L = max(|oc|, r)), and P = o + t d.  With x = a t, the quantity a (|P - c|^2 - r^2) equals
//    (x + hb)^2 - (hb^2 - a (|oc|^2 - r^2)); the computed numerator gives x + hb = -e1 -/+ s with |e1| <= 4u |oc| |d|
//    (rounding of oc and of the dot product) and s the computed square root of the computed discriminant, which
//    differs from the exact one by less than 32u a L^2 (first item of the list above) and is at most |d| L (1 + 2u).
//    Hence | |P - c|^2 - r^2 | <= (8 + 2 + 32) u L^2 <= 48u L^2 after the rounding of the division, i.e.
//    |P - c| <= sqrt(r^2 + 48u L^2): every accepted root is the parameter of a point within
//    e_i = sqrt(r^2 + 48u L^2) - r <= min(0.0017 L, 24u L^2 / r) of the sphere's surface, and so inside the sphere's
//    bounding box grown by e_i on every side.
//  * The walk grows every box by one margin per ray, m = D min(0.003, 26u D / r_min + 2e-6) + 3e-7 max|o_k|, where
//    D >= |o - C| + R for a ball (C, R) around all spheres of the tree, so D >= L for every one of them, and
//    r_min is their smallest radius.  m exceeds every e_i by at least 2e-6 D + 3e-7 max|o_k| (or 0.0013 D).
//  * The slab test computes, per axis, t1 = fma(lo, inv, -fl((o + m) inv)) and t2 = fma(hi, inv, -fl((o - m) inv))
//    with inv = 1/d_k to within 2^-22 (v_rcp_f32), clamped to +-2^90.  Its roundings are those of an exact slab test
//    against planes moved by less than 3u |o_k| + 2^-21 D (the roundings of o +- m, of the products and of inv, the
//    last one acting over at most the 2 D a point of the tree's ball can be away from o); a clamped inv stands for a
//    direction component of 2^-90 instead of a smaller one, which moves a point by less than 2^-47 over the largest
//    parameter an accepted root can have (2 D / |d| <= 2^42).  All of that is inside the slack of the margin, so
//    the computed interval [tn, tf] of a box that contains sphere i contains t.
//  * A child is left out when max(tn, 0) > min(tf, closest).  An accepted root has tmin <= t <= closest (:125-129),
//    so with tn <= t <= tf none of the three reasons (the line misses the box, the box is behind the origin, the
//    box is beyond the closest hit so far) can hold for a box that contains the sphere.  NaN cannot arise: every
//    operand is finite (unused children are boxes at 3e38, beyond every ray of the range, not infinities).
struct QuadBuilder {
    std::vector<Item> items;
    QuadTree* out;
    const rt_gpu_sphere* spheres;

    // Reorders items[lo, hi) so that the `count` first ones are the lowest along the longest axis of the
    // centres' bounding box (ties by sphere index, so that the result does not depend on the input order).
    void split_at(uint32_t lo, uint32_t hi, uint32_t count)
    {
        if (count == 0 || lo + count >= hi) {
            return;
        }
        double mn[3] = { 1e300, 1e300, 1e300 }, mx[3] = { -1e300, -1e300, -1e300 };
        for (uint32_t i = lo; i < hi; ++i) {
            for (int k = 0; k < 3; ++k) {
                mn[k] = std::min(mn[k], items[i].c[k]);
                mx[k] = std::max(mx[k], items[i].c[k]);
            }
        }
        int axis = 0;
        for (int k = 1; k < 3; ++k) {
            if (mx[k] - mn[k] > mx[axis] - mn[axis]) {
                axis = k;
            }
        }
        std::nth_element(items.begin() + lo, items.begin() + lo + count, items.begin() + hi, [&](const Item& a, const Item& b) {
            return a.c[axis] != b.c[axis] ? a.c[axis] < b.c[axis] : a.index < b.index;
        });
    }

    static float round_down(double v)
    {
        const float f = static_cast<float>(v);
        return static_cast<double>(f) > v ? std::nextafterf(f, -INFINITY) : f;
    }
    static float round_up(double v)
    {
        const float f = static_cast<float>(v);
        return static_cast<double>(f) < v ? std::nextafterf(f, INFINITY) : f;
    }

    // Child k of box group `group`: the box around items[lo, hi).
    void set_box(uint32_t group, uint32_t k, uint32_t lo, uint32_t hi)
    {
        double mn[3] = { 1e300, 1e300, 1e300 }, mx[3] = { -1e300, -1e300, -1e300 };
        for (uint32_t i = lo; i < hi; ++i) {
            for (int a = 0; a < 3; ++a) {
                mn[a] = std::min(mn[a], items[i].c[a] - items[i].r);
                mx[a] = std::max(mx[a], items[i].c[a] + items[i].r);
            }
        }
        float* b = out->box.data() + static_cast<size_t>(group) * 24u;
        for (int a = 0; a < 3; ++a) {
            b[4 * a + k] = round_down(mn[a]);
            b[12 + 4 * a + k] = round_up(mx[a]);
        }
    }

    // Last-level group `group` (or the prologue) takes items[lo, hi).
    void fill_leaf_group(uint32_t group, uint32_t lo, uint32_t hi)
    {
        const uint32_t rel = group - out->first_leaf;
        for (uint32_t i = lo; i < hi; ++i) {
            const float* cr = spheres[items[i].index].center_radius;
            const float g[4] = { cr[0], cr[1], cr[2], cr[3] * cr[3] }; // the product the loop over all spheres uses
            std::copy(g, g + 4, out->geom.begin() + (static_cast<size_t>(rel) * 4u + (i - lo)) * 4u);
            out->slot_sphere[4u * rel + (i - lo)] = items[i].index;
        }
    }

    // items[lo, hi) go below group `group` of level `level`.
    void fill_group(uint32_t group, uint32_t level, uint32_t lo, uint32_t hi)
    {
        if (level + 1u == out->levels) {
            fill_leaf_group(group, lo, hi);
            return;
        }
        const uint32_t m = hi - lo;
        uint32_t cut[5];
        cut[0] = lo;
        for (uint32_t k = 0; k < 4u; ++k) {
            cut[k + 1u] = cut[k] + m / 4u + (k < m % 4u ? 1u : 0u);
        }
        split_at(lo, hi, cut[2] - lo);
        split_at(lo, cut[2], cut[1] - lo);
        split_at(cut[2], hi, cut[3] - cut[2]);
        for (uint32_t k = 0; k < 4u; ++k) {
            if (cut[k + 1u] > cut[k]) {
                set_box(group, k, cut[k], cut[k + 1u]);
                fill_group(4u * group + out->roots + k, level + 1u, cut[k], cut[k + 1u]);
            }
        }
    }

    // items[lo, hi) are shared out over the root groups [first, first + count).
    void fill_roots(uint32_t first, uint32_t count, uint32_t lo, uint32_t hi)
    {
        if (count == 1u) {
            fill_group(first, 0u, lo, hi);
            return;
        }
        const uint32_t left = count / 2u;
        const uint32_t m = hi - lo;
        const uint32_t m_left = static_cast<uint32_t>(static_cast<uint64_t>(m) * left / count);
        split_at(lo, hi, m_left);
        fill_roots(first, left, lo, lo + m_left);
        fill_roots(first + left, count - left, lo + m_left, hi);
    }

    // The ball around items[lo, hi) and their smallest radius, for the walk's margin.
    void set_bounds(uint32_t lo, uint32_t hi)
    {
        double mn[3] = { 1e300, 1e300, 1e300 }, mx[3] = { -1e300, -1e300, -1e300 };
        double r_min = 1e300;
        for (uint32_t i = lo; i < hi; ++i) {
            for (int a = 0; a < 3; ++a) {
                mn[a] = std::min(mn[a], items[i].c[a] - items[i].r);
                mx[a] = std::max(mx[a], items[i].c[a] + items[i].r);
            }
            r_min = std::min(r_min, items[i].r);
        }
        double centre[3];
        for (int a = 0; a < 3; ++a) {
            out->bound_centre[a] = static_cast<float>(0.5 * (mn[a] + mx[a]));
            centre[a] = out->bound_centre[a];
        }
        double radius = 0.0;
        for (uint32_t i = lo; i < hi; ++i) {
            double d2 = 0.0;
            for (int a = 0; a < 3; ++a) {
                const double d = items[i].c[a] - centre[a];
                d2 += d * d;
            }
            radius = std::max(radius, std::sqrt(d2) + items[i].r);
        }
        out->bound_radius = round_up(radius * (1.0 + 1e-6));
        out->min_radius = round_down(r_min * (1.0 - 1e-6));
    }
};

} // namespace

bool build_quad_tree(const rt_gpu_sphere* spheres, uint32_t count, QuadTree& out)
{
    out.clear();
    QuadBuilder b;
    b.out = &out;
    b.spheres = spheres;
    std::vector<Item>& items = b.items;
    if (!collect_items(spheres, count, items)) {
        return false;
    }
    // Spheres more than four times the median radius: the (up to four) largest go into the prologue.  Measured on
    // the random scene (485 spheres, three of radius 1 among spheres of radius 0.2, and the ground): with a factor of
    // sixteen only the ground is kept out and the three large spheres widen the boxes above them, 13.6 ms; with four
    // they are tested with the ground at the start of every segment, 11.9 ms.
    std::vector<double> radii(count);
    for (uint32_t i = 0; i < count; ++i) {
        radii[i] = items[i].r;
    }
    std::nth_element(radii.begin(), radii.begin() + count / 2u, radii.end());
    const double big = 4.0 * radii[count / 2u];
    std::stable_sort(items.begin(), items.end(), [&](const Item& x, const Item& y) {
        const bool bx = x.r > big, by = y.r > big;
        return bx != by ? bx : (bx && x.r != y.r ? x.r > y.r : false);
    });
    uint32_t outliers = 0;
    while (outliers < 4u && outliers < count && items[outliers].r > big) {
        ++outliers;
    }
    const uint32_t in_tree = count - outliers;
    if (in_tree == 0) {
        return false;
    }
    // Shape: the fewest groups on the last level for up to kQuadMaxRoots roots; of equal ones the fewer roots.
    const uint64_t leaf_groups_needed = (static_cast<uint64_t>(in_tree) + 3u) / 4u;
    uint64_t best_leaf_groups = 0;
    uint64_t per_root = 1; // groups of the last level below one root
    for (uint32_t levels = 1; levels <= kQuadMaxLevels + 1u; ++levels, per_root *= 4u) {
        const uint64_t roots = (leaf_groups_needed + per_root - 1u) / per_root;
        if (roots > kQuadMaxRoots) {
            continue;
        }
        const uint64_t leaf_groups = roots * per_root;
        if (out.levels == 0 || leaf_groups < best_leaf_groups || (leaf_groups == best_leaf_groups && roots < out.roots)) {
            out.levels = levels;
            out.roots = static_cast<uint32_t>(roots);
            best_leaf_groups = leaf_groups;
        }
        if (roots <= 1u) {
            break;
        }
    }
    if (out.levels == 0) {
        out.clear();
        return false;
    }
    uint64_t per_level = out.roots, tree_groups = 0, first_leaf = 0;
    for (uint32_t l = 0; l < out.levels; ++l, per_level *= 4u) {
        first_leaf = tree_groups;
        tree_groups += per_level;
    }
    const uint64_t sphere_groups = tree_groups - first_leaf + 1u; // the last level and the prologue
    out.first_leaf = static_cast<uint32_t>(first_leaf);
    out.prologue_spheres = outliers;
    out.box.assign(first_leaf * 24u, kQuadUnusedBox);
    const float unused[4] = { 0.0f, 0.0f, 0.0f, -INFINITY };
    out.geom.resize(sphere_groups * 16u);
    for (size_t slot = 0; slot < out.geom.size() / 4u; ++slot) {
        std::copy(unused, unused + 4, out.geom.begin() + slot * 4u);
    }
    out.slot_sphere.assign(sphere_groups * 4u, kTreeNoSphere);
    b.fill_leaf_group(static_cast<uint32_t>(tree_groups), 0u, outliers); // the prologue sits behind the tree
    b.fill_roots(0u, out.roots, outliers, count);
    b.set_bounds(outliers, count);
    return true;
}

bool build_sphere_tree(const rt_gpu_sphere* spheres, uint32_t count, SphereTree& out)
{
    out.clear();
    Builder b;
    b.out = &out;
    if (!collect_items(spheres, count, b.items)) {
        return false;
    }
    b.emit_children(0, count, spheres);
    const uint32_t nodes = out.node_count();
    if (nodes <= kTreeMaxNodes && count <= (kTreeLeafFlag >> kTreeIndexShift)) {
        out.word.resize(nodes);
        for (uint32_t i = 0; i < nodes; ++i) {
            const uint32_t leaf = out.sphere[i] == kTreeNoSphere ? 0u : kTreeLeafFlag | (out.sphere[i] << kTreeIndexShift);
            out.word[i] = leaf | (out.link[i] * 16u);
        }
    }
    return true;
}

} // namespace rt

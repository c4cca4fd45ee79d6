// Hierarchies over a sphere list, for scenes where testing every sphere against every ray segment
// (the loop of shaders/raytrace.comp.glsl:247-259) dominates the run time: a depth-first tree of
// bounding balls (SphereTree, described first; walked on request only) and the complete four-way tree
// of bounding boxes that the tracer uses (QuadTree, further down).
//
// The hierarchy only decides WHICH spheres a ray is tested against; every sphere that is tested
// goes through exactly the arithmetic of hitSphere (:109-133), and the closest-hit rule of
// :254-258 (smallest accepted root, the later sphere winning an exact tie) does not depend on the
// order of the tests.  The result is therefore identical to the loop over all spheres as long as
// no sphere is skipped that the loop would have accepted; the inflation applied to the bounding
// spheres below is what guarantees that, including for discriminants that only rounding makes
// non-negative.
//
// Layout: one array of nodes in depth-first order.  A node is a float4 (centre, w) plus one
// 32-bit word:
//   leaf      w = radius^2 (the float product the brute-force loop uses)
//             word = 1<<31 | sphere index << 17 | byte offset of the next node
//   internal  w = 1.125 * R^2 rounded up, R = radius of a ball around `centre` that contains all
//             spheres below the node;  word = byte offset (index * 16) of the node that follows
//             the whole subtree
// A ray visits node i; if the node's test fails it continues at the offset in the word, otherwise
// at i + 1 (the first child, or simply the next node after a leaf).
//
// (The copy in device memory of such a tree also carries an inner node's continuation index in the low
// 13 mantissa bits of w, rounded up first, with the sign bit set: rt_host.cpp, tree_walk.)
//
// Trees too large for those bit fields (more than kTreeMaxNodes nodes) keep two words per node
// instead: the byte offset of the node to continue at, and 1<<31 | sphere index (0 for an internal
// node); such trees are walked in global memory.
#pragma once

#include <stdint.h>

#include <vector>

#include "../../include/rt_mi355x.h"

namespace rt {

// Scenes smaller than this are traced by the plain loop.  Measured on MI355X with the synthetic grid
// scene (960 x 540, 8 spp, depth 16, 4 frames; loop / hierarchy in ms): 8 spheres 1.20 / 1.35,
// 12: 1.58 / 1.45, 16: 1.77 / 1.52, 24: 2.22 / 1.94, 32: 2.89 / 1.98, 64: 4.59 / 2.57.  Scenes whose
// spheres all overlap, are nested or stand in a row seen end-on gain nothing from the walk: measured
// between 15 % faster and 50 % slower than the loop at 24 to 200 spheres, hence not the break-even
// point itself.
#ifndef RT_TREE_MIN_SPHERES
#define RT_TREE_MIN_SPHERES 24
#endif
constexpr uint32_t kTreeMinSpheres = RT_TREE_MIN_SPHERES;
constexpr uint32_t kTreeLeafFlag = 0x80000000u;
constexpr uint32_t kTreeOffsetMask = 0x0001FFFFu;
constexpr uint32_t kTreeIndexShift = 17;
// Most nodes a tree in the packed format may have: offsets are 17 bits of bytes / 16 (8191), and
// 20 bytes per node plus the hash tables of sixteen waves have to fit the 160 KiB of LDS of a
// compute unit.
constexpr uint32_t kTreeMaxNodes = 7850;

// Largest scene a hierarchy is built for (byte offsets of nodes stay below 2^32 with room to spare).
constexpr uint32_t kTreeMaxSpheres = 1u << 20;
constexpr uint32_t kTreeNoSphere = 0xFFFFFFFFu;

struct SphereTree {
    std::vector<float> geom;       // 4 floats per node
    std::vector<uint32_t> link;    // per node: index of the node to continue at when the node's test fails
    std::vector<uint32_t> sphere;  // per node: sphere index of a leaf, kTreeNoSphere for an internal node
    std::vector<uint32_t> word;    // per node: the packed word; empty when the tree does not fit the bit fields
    uint32_t node_count() const { return static_cast<uint32_t>(link.size()); }
    bool empty() const { return link.empty(); }
    bool packed() const { return !word.empty(); }
    void clear()
    {
        geom.clear();
        link.clear();
        sphere.clear();
        word.clear();
    }
};

// ---- a complete four-way tree of bounding boxes (kept in LDS when it fits, in global memory otherwise) -----
// Groups of four children, visited one group per trip of the walk: the four tests are independent, so their LDS
// reads are in flight together and a segment's chain of dependent reads is short.  The tree is complete and
// implicit: `roots` groups on level 0, four times as many on each level below, `levels` levels; group g of the
// tree has the children 4 g + roots + k (k = 0..3) and the parent (g - roots) / 4, so no links are stored.
//
// A group above the last level holds the axis-aligned bounding boxes of its four child groups, 24 floats laid
// out as lo.x[4], lo.y[4], lo.z[4], hi.x[4], hi.y[4], hi.z[4]: the smallest box around the spheres below the
// child, each bound rounded outward to binary32.  Boxes are far tighter than balls around spheres that lie on
// a plane (a ray that passes ten radii above a field of small spheres misses the box of every part of it).
// An unused child has lo = hi = 3e38 on every axis, which no ray of the bounded range reaches.
// A group of the last level holds four spheres, (centre, radius^2), tested with hitSphere's own arithmetic;
// an unused slot has w = -infinity, which no ray passes.  Spheres are dealt out evenly (every split is at the
// median count, along the longest axis of the part's centres), so the depth of every sphere is the same.
// Spheres far larger than the rest (the ground under a field of small spheres) would inflate every box above
// them; up to four of them are kept out of the tree in one more group of the last-level kind, the prologue,
// stored behind the tree (group index group_count - 1) and tested first for every ray.
// slot_sphere[4 (g - first_leaf) + k] is the sphere in slot k of last-level group g (the prologue included).
//
// The walk inflates every box by a margin that depends on the ray (rt_tree.cpp derives it); for that it needs a
// ball (bound_centre, bound_radius) around all spheres of the tree (the prologue's excluded) and their smallest
// radius.
struct QuadTree {
    uint32_t levels = 0;      // 0: no tree
    uint32_t roots = 0;       // groups on level 0, 1..kQuadMaxRoots
    uint32_t first_leaf = 0;  // index of the first group of the last level = number of box groups
    uint32_t prologue_spheres = 0; // spheres in the prologue, 0..4 (they fill its first slots)
    float bound_centre[3] = { 0.0f, 0.0f, 0.0f };
    float bound_radius = 0.0f;
    float min_radius = 0.0f;
    std::vector<float> box;            // 24 floats per group above the last level
    std::vector<float> geom;           // 16 floats per group of the last level, then the prologue
    std::vector<uint32_t> slot_sphere; // kTreeNoSphere for an unused slot
    uint32_t group_count() const { return levels == 0 ? 0u : first_leaf + static_cast<uint32_t>(geom.size() / 16u); }
    uint32_t slot_count() const { return static_cast<uint32_t>(slot_sphere.size()); }
    bool has_prologue() const { return prologue_spheres != 0; }
    // bytes of the tree as the kernel keeps it in LDS: 96 per box group, 64 per group of spheres
    size_t lds_bytes() const { return static_cast<size_t>(first_leaf) * 96u + (geom.size() / 16u) * 64u; }
    bool empty() const { return levels == 0; }
    void clear()
    {
        levels = roots = first_leaf = prologue_spheres = 0;
        bound_radius = min_radius = 0.0f;
        box.clear();
        geom.clear();
        slot_sphere.clear();
    }
};
constexpr uint32_t kQuadMaxRoots = 16;
// Most bytes a four-way tree kept in LDS may take: with the hash tables of sixteen waves it has to fit the
// 160 KiB of a compute unit.  Larger trees are walked in global memory by the same code.
constexpr size_t kQuadMaxLdsBytes = 152u * 1024u;
// Levels below the roots the walk's stack of pending-sibling masks (four bits per level in one 32-bit register) holds.
// Sixteen roots and eight levels below them reach 2^20 groups of the last level, four times the largest scene.
constexpr uint32_t kQuadMaxLevels = 8;
constexpr float kQuadUnusedBox = 3.0e38f;

// Builds the four-way tree for the scenes build_sphere_tree accepts; returns false (and leaves `out` empty) for the
// others.  out.lds_bytes() says whether the tree fits LDS (kQuadMaxLdsBytes).
bool build_quad_tree(const rt_gpu_sphere* spheres, uint32_t count, QuadTree& out);

// Builds the hierarchy.  Returns false (and leaves `out` empty) when the scene does not qualify:
// fewer than kTreeMinSpheres or more than kTreeMaxSpheres spheres, or any centre/radius outside the
// ranges the error bounds were derived for (non-finite values, |centre| > 2^20, radius^2 outside
// [2^-40, 2^40]).
bool build_sphere_tree(const rt_gpu_sphere* spheres, uint32_t count, SphereTree& out);

} // namespace rt

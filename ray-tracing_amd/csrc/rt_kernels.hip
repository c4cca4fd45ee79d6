// HIP kernels of the sphere path tracer for gfx950 (MI355X).
//
// rt_trace_kernel is the counterpart of main() in shaders/raytrace.comp.glsl:298-343,
// restructured for 64-wide wavefronts.
//
// Work decomposition.  The unit of independent work is an ITEM = one (pixel, frame) pair: its
// `spp` samples share one random stream (:310-331) whose draw count per sample is data dependent,
// so they stay on one lane, in order.  A wave owns a UNIT = a small pixel tile (2x4 to 8x16, chosen
// per launch) x a chunk of up to `frames_per_unit` frames.  Lanes pull items from that pool with a
// ballot/prefix count (no atomics, no LDS): a lane that finishes its item takes the next one, so
// all 64 lanes stay busy until the pool drains, whatever the path lengths were.
//
// The loop body is one RAY SEGMENT per lane: the bounce loop (:242-293) and the sample loop
// (:315-331) are flattened, and a lane whose path ended generates its next camera ray in the same
// iteration.  The five normalize() calls of the shader (camera :328, lambert :72 and :159, metal
// :201, dielectric :215, sky :263) are merged into two per iteration with per-lane selected
// operands, and the rejection samplers of lambert (:157) and metal (:204) share one loop, so the
// expensive correctly rounded sqrt/divide sequences run with most lanes active.
//
// Accumulation.  A launch of one frame adds the frame's sum to the accumulator texel directly
// (:333-335).  A launch of several frames stores each item's sum in a per-frame image
// (partial[frame][pixel], HBM) and rt_fold_kernel then adds them to the texel in frame order, which
// reproduces the float additions of consecutive dispatches exactly.
//
// The sphere list is staged once per workgroup in LDS as (centre, radius^2) float4 records read at
// a wave-uniform address (broadcast); material records are fetched only for the winning sphere.
// Scenes of up to 16 spheres: iterations in which every ray of the wave is a camera ray of the wave's
// own pixel tile test only the spheres that tile can see; the kernel for exactly five spheres (the
// reference's scene) has its sphere loop unrolled.
//
// Scenes of kTreeMinSpheres spheres or more are traced through a hierarchy (rt_tree.hpp): a complete
// four-way tree of bounding boxes that each lane walks on its own, four children per trip (quad_walk);
// a sphere it reaches goes through the same arithmetic as in the plain loop.  A lane's walk may stretch
// over several iterations of the trace loop, and the waves of that kernel take their units from a
// counter in memory.  Trees too large for LDS are walked in global memory.  (tree_walk walks the older
// depth-first tree of bounding balls, one node per trip, on request.)
//
// Compile with -ffp-contract=off (see rt_device.hpp).
#include "rt_device.hpp"
#include "rt_kernels.hpp"
#include "rt_tree.hpp"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <mutex>

namespace rt {

namespace {

#ifndef RT_WAVES_PER_SIMD
#define RT_WAVES_PER_SIMD 1
#endif
// Waves per SIMD asked of the compiler for the walks of the four-way tree in workgroups of up to 512 threads (1 = no
// constraint: 103 vector registers, four waves).  With 5 the kernels fit 96 registers and a fifth wave, at 8 to 24
// bytes of spilled registers per lane: within 1 % either way on the 485-sphere scene and on grids of 1024 to 262 144
// spheres (the walk is bound by instruction issue, not by latency).
#ifndef RT_QUAD_WAVES_PER_SIMD
#define RT_QUAD_WAVES_PER_SIMD 1
#endif
// Scalar registers per wave.  96 is the most that lets eight waves share a SIMD (the kernel needs
// fewer than 64 vector registers); left to itself the compiler takes about 104, which caps the
// SIMD at seven waves and costs 4.6 % on the benchmark.
#ifndef RT_NUM_SGPR
#define RT_NUM_SGPR 96
#endif
#define RT_SGPR_ATTR __attribute__((amdgpu_num_sgpr(RT_NUM_SGPR)))
constexpr int kBlockThreads = 256;
constexpr float kTMin = 0.001f;
// Lanes that must want a new camera sample before any of them gets one (0 = no waiting).
#ifndef RT_REGEN_QUORUM
#define RT_REGEN_QUORUM 16
#endif
constexpr uint32_t kRegenQuorum = RT_REGEN_QUORUM;
// Lanes with something else to do that end a round of the hierarchy walk (0 = walk until every lane is done).
#ifndef RT_WALK_QUORUM
#define RT_WALK_QUORUM 24
#endif
constexpr uint32_t kWalkQuorum = RT_WALK_QUORUM;
// The same for the walk of the four-way tree, whose trips are cheap next to the shading blocks that follow a round.
// Measured on the 4096-sphere scene (1920 x 1080, depth 50, 8 frames): 12: 48.5 ms, 16: 45.5, 24: 41.9, 32: 40.9,
// 40: 40.7, 48: 41.5, 56: 44.6 (with a fixed quorum of 32 lanes for the groups of spheres).  Again once a box trip
// had lost a quarter of its instructions (no per-axis minimum and maximum), 8 frames: 32: 34.1 ms, 40: 32.0, 44: 31.5,
// 48: 31.2, 52: 31.2, 56: 31.6, 60: 32.6.
#ifndef RT_QUAD_WALK_QUORUM
#define RT_QUAD_WALK_QUORUM 48
#endif
constexpr uint32_t kQuadWalkQuorum = RT_QUAD_WALK_QUORUM;
// Four-way tree: the groups of spheres of the last level are tested when this share (in eighths) of the lanes that
// are walking stand at one; 0 = in every trip.  Measured as above with a walk quorum of 40: every trip 49.2 ms;
// fixed quorums of 16 / 24 / 32 / 40 lanes 43.3 / 42.1 / 41.9 / 43.6 ms (walk quorum 24); shares of 3 / 4 / 5 eighths
// 38.2 / 38.2 / 38.0 ms.
#ifndef RT_LEAF_SHARE
#define RT_LEAF_SHARE 4
#endif
constexpr uint32_t kLeafShare = RT_LEAF_SHARE;

// ---- diagnostic section counters -------------------------------------------------------------
// A diagnostic build (-DRT_STATS) counts, per code section, how often a wave executes it and the
// number of active lanes.  The shipped library instantiates the kernel with NoStats only.
enum Section {
    ST_ITER, ST_ITEM, ST_GEN, ST_DISK_IT, ST_NORM2, ST_ROOT, ST_FAR_ROOT, ST_HIT, ST_RIS, ST_RIS_IT,
    ST_SKY, ST_LAMBERT, ST_CHECKER, ST_METAL, ST_DIEL, ST_SCHLICK, ST_NODE, ST_ALL_PRIMARY, ST_SPHERE_TEST, ST_COUNT
};

enum Phase { PH_NORM2, PH_INTERSECT, PH_HIT, PH_ITEM, PH_SAMPLER, PH_NORM1, PH_SHADE, PH_TAIL, PH_COUNT };

struct NoStats {
    [[maybe_unused]] static constexpr bool kEnabled = false;
    __device__ __forceinline__ void section(int) {}
    __device__ __forceinline__ void phase_start() {}
    __device__ __forceinline__ void phase_end(int) {}
    __device__ __forceinline__ void flush(uint32_t*) {}
};

struct WaveStats {
    [[maybe_unused]] static constexpr bool kEnabled = true;
    uint32_t exec[ST_COUNT];
    uint32_t lanes[ST_COUNT];
    unsigned long long phase_cycles[PH_COUNT]; // shader-clock cycles between phase marks (wave-uniform)
    unsigned long long last;
    __device__ __forceinline__ WaveStats()
    {
        for (int i = 0; i < ST_COUNT; ++i) {
            exec[i] = 0;
            lanes[i] = 0;
        }
        for (int i = 0; i < PH_COUNT; ++i) {
            phase_cycles[i] = 0;
        }
        last = 0;
    }
    static __device__ __forceinline__ unsigned long long stamp()
    {
        unsigned long long t;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        return t;
    }
    __device__ __forceinline__ void phase_start() { last = stamp(); }
    __device__ __forceinline__ void phase_end(int k)
    {
        unsigned long long t = stamp();
        phase_cycles[k] += t - last;
        last = t;
    }
    __device__ __forceinline__ void section(int k)
    {
        unsigned long long m = __builtin_amdgcn_ballot_w64(true);
        lanes[k] += 1u;
        if ((threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)m) - 1)) {
            exec[k] += 1u;
        }
    }
    __device__ __forceinline__ void flush(uint32_t* out)
    {
        if (out == nullptr) {
            return;
        }
        for (int i = 0; i < ST_COUNT; ++i) {
            if (exec[i]) {
                atomicAdd(reinterpret_cast<unsigned long long*>(out) + 2 * i, (unsigned long long)exec[i]);
            }
            if (lanes[i]) {
                atomicAdd(reinterpret_cast<unsigned long long*>(out) + 2 * i + 1, (unsigned long long)lanes[i]);
            }
        }
        if ((threadIdx.x & 63u) == 0u) {
            for (int i = 0; i < PH_COUNT; ++i) {
                atomicAdd(reinterpret_cast<unsigned long long*>(out) + 2 * ST_COUNT + i, phase_cycles[i]);
            }
        }
    }
};

// How the sphere list reaches the intersection loop.
enum Persistence { kOneUnit = 0, kByCounter = 1 };

enum SceneMode {
    kGeomLds = 0, // geometry records in LDS, material records read from memory for the winner
    kFullLds = 1, // whole records in LDS
    kTree = 2,    // hierarchy nodes in LDS, sphere records read from memory for the winner
    kTreeGlobal = 3, // hierarchy too large for LDS (or for the packed node word): walked in global memory
    kGeomGlobal = 4, // no hierarchy and too many spheres for LDS: geometry records read from global memory
    kTreeQuad = 6,   // the complete four-way tree of boxes (rt_tree.hpp: QuadTree) in LDS, four children per trip of the walk
    kTreeQuadGlobal = 7, // the same tree, too large for LDS, walked in global memory
    kFullLds5 = 5    // kFullLds for exactly five spheres (the reference's own scene, src/rt/RayTracer.cpp:105-138):
                     // the sphere loop is unrolled and every LDS offset is a constant
};

// Copies the scene into LDS: geometry always, material records when they fit.
template <bool MAT_IN_LDS>
__device__ __forceinline__ void stage_scene(const rt_gpu_sphere* __restrict__ spheres, uint32_t count,
                                            float4* geom, float4* mat0, float4* mat1, float4* hit_rec)
{
    for (uint32_t i = threadIdx.x; i < count; i += blockDim.x) {
        const float4* src = reinterpret_cast<const float4*>(spheres + i);
        float4 cr = src[0];
        geom[i] = make_float4(cr.x, cr.y, cr.z, cr.w * cr.w);
        if (MAT_IN_LDS) {
            float4 al = src[1];
            mat0[i] = make_float4(al.x, al.y, al.z, cr.w);
            // material code as the integer of :145; the slot of the fuzz (read by metal only) holds
            // 1 / refIdx (:218) for every other material
            const float4 m = src[2];
            const uint32_t code = (uint32_t)(m.x + 0.5f);
            mat1[i] = make_float4(__uint_as_float(code), code == 1u ? m.y : 1.0f / m.z, m.z, m.w);
            // centre and the reciprocal of the radius as the normal's division uses it (NaN: radius
            // outside the range of the short division sequence)
            hit_rec[i] = make_float4(cr.x, cr.y, cr.z, den_in_range(cr.w) ? refined_rcp(cr.w) : __builtin_nanf(""));
        }
    }
    __syncthreads();
}

// Copies the hierarchy into LDS.
__device__ __forceinline__ void stage_tree(const float4* __restrict__ node_geom, const uint32_t* __restrict__ node_word,
                                           uint32_t count, float4* geom, uint32_t* word)
{
    for (uint32_t i = threadIdx.x; i < count; i += blockDim.x) {
        geom[i] = node_geom[i];
        word[i] = node_word[i];
    }
    __syncthreads();
}

// Copies the four-way tree into LDS: six float4 per box group, then four per group of spheres.
__device__ __forceinline__ void stage_quad_tree(const float4* __restrict__ tree, uint32_t records, float4* lds_tree)
{
    for (uint32_t i = threadIdx.x; i < records; i += blockDim.x) {
        lds_tree[i] = tree[i];
    }
    __syncthreads();
}

// Mask of the lanes for which `predicate` holds.  (__ballot takes an int: the predicate would be turned
// into 0/1 by a select and compared with zero again, two vector instructions per ballot.)
__device__ __forceinline__ unsigned long long wave_ballot(bool predicate)
{
    return __builtin_amdgcn_ballot_w64(predicate);
}

// Number of set bits of a wave mask below this lane.
__device__ __forceinline__ uint32_t lane_prefix(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// The kernel's argument struct as it lies in the kernel-argument segment (it is the only argument).
// Blocks that run rarely read what they need from there instead of keeping it in scalar registers
// through the whole trace loop, which is short of them; the empty asm keeps the compiler from
// recognising the copy.  The pointer stays in the constant address space, so that the reads are scalar
// loads (through a generic pointer they would be per-lane flat loads).
typedef const __attribute__((address_space(4))) TraceArgs* KernelArgs;
__device__ __forceinline__ KernelArgs args_in_memory()
{
    auto segment = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(segment));
    return (KernelArgs)segment;
}

// Closest hit of one ray segment over the whole scene, raytrace.comp.glsl:244-259: `closest` and
// `best` come back as the loop of :247-259 leaves them (1e20 and -1 without a hit).
template <int MODE, class Stats>
__device__ __forceinline__ void closest_hit(const TraceArgs& A, const float4* geom, const uint32_t* node_word, uint32_t sphere_count,
                                            uint32_t sphere_mask, f3 ro, f3 rd, RayDenominator& rden, float& closest, int& best,
                                            Stats& st)
{
    constexpr bool MAT_IN_LDS = MODE == kFullLds || MODE == kFullLds5;
    closest = 1e20f;
    best = -1;
    rden = ray_denominator(ro, rd, A.scene_bounded != 0u);
    if (MODE == kFullLds5) {
        // unrolled, a scalar branch per sphere; the mask is 31 unless every ray is a camera ray
#pragma unroll
        for (uint32_t i = 0; i < 5u; ++i) {
            if ((sphere_mask & (1u << i)) != 0u) {
                st.section(ST_SPHERE_TEST);
                sphere_closest(sphere_quadratic(geom[i], ro, rd, rden.a), rden, kTMin, (int)i, closest, best, st);
            }
        }
        return;
    }
    if (sphere_mask != 0xFFFFFFFFu) {
        // small scenes: the spheres of the mask, in index order
        for (uint32_t m = sphere_mask; m != 0u; m &= m - 1u) {
            const uint32_t i = (uint32_t)__builtin_ctz(m);
            st.section(ST_SPHERE_TEST);
            sphere_closest(sphere_quadratic(geom[i], ro, rd, rden.a), rden, kTMin, (int)i, closest, best, st);
        }
        return;
    }
    uint32_t i = 0;
    if (!MAT_IN_LDS) {
        // Large scenes: four records per trip, fetched together so that one LDS wait covers
        // them and the four discriminants overlap; the roots are still resolved in index
        // order (an equal-t later sphere replaces an earlier one, :125 with tMax = closest).
        for (; i + 4u <= sphere_count; i += 4u) {
            const float4 g0 = geom[i], g1 = geom[i + 1u], g2 = geom[i + 2u], g3 = geom[i + 3u];
            const SphereQuadratic q0 = sphere_quadratic(g0, ro, rd, rden.a);
            const SphereQuadratic q1 = sphere_quadratic(g1, ro, rd, rden.a);
            const SphereQuadratic q2 = sphere_quadratic(g2, ro, rd, rden.a);
            const SphereQuadratic q3 = sphere_quadratic(g3, ro, rd, rden.a);
            // one branch per four spheres: almost always all four discriminants are negative
            if (!(q0.discriminant < 0.0f) || !(q1.discriminant < 0.0f) || !(q2.discriminant < 0.0f) ||
                !(q3.discriminant < 0.0f)) {
                float root;
                if (sphere_root_from(q0, rden, kTMin, closest, root, st)) {
                    closest = root;
                    best = (int)i;
                }
                if (sphere_root_from(q1, rden, kTMin, closest, root, st)) {
                    closest = root;
                    best = (int)i + 1;
                }
                if (sphere_root_from(q2, rden, kTMin, closest, root, st)) {
                    closest = root;
                    best = (int)i + 2;
                }
                if (sphere_root_from(q3, rden, kTMin, closest, root, st)) {
                    closest = root;
                    best = (int)i + 3;
                }
            }
        }
    }
    for (; i < sphere_count; ++i) {
        sphere_closest(sphere_quadratic(geom[i], ro, rd, rden.a), rden, kTMin, (int)i, closest, best, st);
    }
}

// Walk of the hierarchy, resumable: a lane's search for the closest hit of its segment can stretch
// over several iterations of the trace loop.  Lanes need walks of very different lengths (66 to 144
// nodes on the 4096-sphere grid), so instead of letting the whole wave wait for its longest walk the
// walk loop stops as soon as kWalkQuorum lanes have something else to do (their walk is over, or they
// carry no ray); those lanes are shaded and given new rays while the others keep their place.
struct TreeWalk {
    uint32_t at;         // byte offset of this lane's node; at or past the end of the array: no walk in progress
    float closest;
    int best;
    float beyond_limit;  // 2.5 * dot(d,d) * closest^2
    RayDenominator rden;
    float4 g;            // the node at `at`, fetched ahead
    uint32_t w;
    uint32_t link;       // byte offset of the node to continue at when this node's test fails
    // the four-way tree instead of `at`, `g`, `w`, `link`, `beyond_limit`:
    uint32_t grp;        // this lane's group; kQuadIdle: no walk in progress
    uint32_t stack;      // four bits per level above `grp`: the siblings on that level that passed their parent's
                         // test and have not been visited yet (lowest four bits: the siblings of `grp` itself)
    uint32_t root;       // the level-0 group the walk is below
    f3 inv;              // 1 / d per axis, clamped to +-2^90
    f3 near_off, far_off; // per axis -(o + m) * inv and -(o - m) * inv, m the margin of this ray's boxes (rt_tree.cpp):
                         // the first belongs to the plane the ray crosses first, i.e. they are swapped where inv < 0
    uint32_t sel_x, sel_y, sel_z; // 48 where inv < 0 on that axis, else 0: byte offset from a group's low bounds to its high ones
};
constexpr uint32_t kQuadIdle = 0xFFFFFFFFu;

// One node's words: packed (one word, rt_tree.hpp) or wide (continuation offset, then leaf flag | sphere index).
template <bool WIDE>
__device__ __forceinline__ void load_node_words(const uint32_t* node_word, uint32_t at, uint32_t& w, uint32_t& link)
{
    if (WIDE) {
        const uint2 ww = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(node_word) + (at >> 1));
        link = ww.x;
        w = ww.y;
    } else {
        w = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(node_word) + (at >> 2));
        link = w & kTreeOffsetMask;
    }
}

template <bool WIDE, class Stats>
__device__ __forceinline__ void tree_walk(const TraceArgs& A, const float4* geom, const uint32_t* node_word, uint32_t sphere_count,
                                          bool live, f3 ro, f3 rd, TreeWalk& t, Stats& st)
{
    const uint32_t end = A.tree_nodes * 16u;
    if (live && t.at >= end) {
        // a new segment
        t.closest = 1e20f;
        t.best = -1;
        t.beyond_limit = __builtin_inff();
        t.rden = ray_denominator(ro, rd, true); // a scene with a hierarchy is bounded
        // the skip tests of the hierarchy are proven for rays in the bounded range only
        if (t.rden.bounded) {
            t.at = 0;
            t.g = geom[0];
            load_node_words<WIDE>(node_word, 0u, t.w, t.link);
        } else {
            RT_COLD_PATH();
            for (uint32_t i = 0; i < sphere_count; ++i) {
                const float4 cr = *reinterpret_cast<const float4*>(A.spheres + i);
                const float4 g = make_float4(cr.x, cr.y, cr.z, cr.w * cr.w);
                sphere_closest(sphere_quadratic(g, ro, rd, t.rden.a), t.rden, kTMin, (int)i, t.closest, t.best, st);
            }
        }
    }
    // the round ends when fewer than this many lanes are still walking
    const uint32_t in_loop = (uint32_t)__popcll(wave_ballot(true));
    const uint32_t few = __builtin_amdgcn_readfirstlane(kWalkQuorum != 0u && in_loop > kWalkQuorum ? in_loop - kWalkQuorum : 0u);
    if ((uint32_t)__popcll(wave_ballot(t.at < end)) <= few) {
        return;
    }
    // (bottom-tested by hand: the compiler does not rotate a loop whose test holds a ballot)
    do {
        if (t.at < end) {
            st.section(ST_NODE);
            // The node after this one is either the next in the array or the one behind this node's
            // subtree (for a leaf the two coincide); both are fetched now, so that the memory latency
            // passes while this node is tested.  Behind the last node both offsets point at the first
            // bytes after the arrays they index (inside the workgroup's LDS, or the spare entry of the
            // arrays in global memory), and what is read there is not used.
            const uint32_t seq = t.at + 16u;
            if (WIDE) {
                const float4 g = t.g;
                const uint32_t w = t.w;
                const uint32_t jump = t.link;
                const float4 g_seq = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(geom) + seq);
                const float4 g_jump = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(geom) + jump);
                uint32_t w_seq, link_seq, w_jump, link_jump;
                load_node_words<WIDE>(node_word, seq, w_seq, link_seq);
                load_node_words<WIDE>(node_word, jump, w_jump, link_jump);
                // dot(oc,oc) is scaled by 1 - 2^-13 for an internal node and by exactly 1 for a leaf
                const float k = __uint_as_float(0x3F7FF800u + ((w >> 20) & 0x800u));
                const SphereQuadratic q = sphere_quadratic_scaled(g, ro, rd, t.rden.a, k);
                // A ball (radius R, centre at distance D from the origin) lies wholly beyond the closest
                // hit so far when D > L + R, L = closest * |d|; D^2 - 2 R^2 > 2.5 L^2 implies D > 1.05 (L + R),
                // which no rounding here comes near, so every root inside the ball exceeds `closest`.
                const bool skip = q.discriminant < 0.0f || behind_origin(q.half_b, q.c) || q.c - g.w > t.beyond_limit;
                if (!skip && (int32_t)w < 0) {
                    sphere_closest_any_order(q, t.rden, kTMin, (int)(w & ~kTreeLeafFlag), t.closest, t.best, st);
                    t.beyond_limit = 2.5f * t.rden.a * t.closest * t.closest;
                }
                t.g = skip ? g_jump : g_seq;
                t.w = skip ? w_jump : w_seq;
                t.link = skip ? link_jump : link_seq;
                t.at = skip ? jump : seq;
            } else {
                // In LDS the fourth float of an inner node carries its sign bit set and, in the low 13 bits
                // of its mantissa, the index of the node to continue at (rt_host.cpp packs it; the value
                // is w rounded up, so the test only gets more cautious); a leaf's is r^2 as it is.  The
                // word array is read for the sphere index of a leaf that has to be resolved, nothing else.
                const uint32_t bits = __float_as_uint(t.g.w);
                const bool inner = (int32_t)bits < 0;
                const float4 g = make_float4(t.g.x, t.g.y, t.g.z, __uint_as_float(bits & 0x7FFFFFFFu));
                const uint32_t jump = inner ? (bits & 0x1FFFu) << 4 : seq;
                const float4 g_seq = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(geom) + seq);
                const float4 g_jump = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(geom) + jump);
                const float k = __uint_as_float(0x3F800000u - ((bits >> 20) & 0x800u));
                const SphereQuadratic q = sphere_quadratic_scaled(g, ro, rd, t.rden.a, k);
                const bool skip = q.discriminant < 0.0f || behind_origin(q.half_b, q.c) || q.c - g.w > t.beyond_limit;
                if (!skip && !inner) {
                    const uint32_t w = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(node_word) + (t.at >> 2));
                    sphere_closest_any_order(q, t.rden, kTMin, (int)((w & ~kTreeLeafFlag) >> kTreeIndexShift), t.closest, t.best, st);
                    t.beyond_limit = 2.5f * t.rden.a * t.closest * t.closest;
                }
                t.g = skip ? g_jump : g_seq;
                t.at = skip ? jump : seq;
            }
        }
    } while ((uint32_t)__popcll(wave_ballot(t.at < end)) > few);
}

// The sphere index kept with slot `slot` of the sphere list in slot order (TraceArgs::spheres of the four-way tree).
__device__ __forceinline__ uint32_t slot_sphere_index(const rt_gpu_sphere* slot_spheres, int slot)
{
    return __float_as_uint(slot_spheres[slot].albedo[3]);
}

// sphere_closest_any_order for the four-way tree, where a sphere is known by its slot: the smaller root wins, and
// of exactly equal roots the higher SPHERE index (:125 does not reject root == closest, and the shader's loop runs in
// index order), which is looked up only when such a tie occurs.
template <class Stats>
__device__ __forceinline__ void sphere_closest_slot(const SphereQuadratic& q, const RayDenominator& rden, float tmin, int slot,
                                                    const rt_gpu_sphere* slot_spheres, float& closest, int& best, Stats& st)
{
    st.section(ST_ROOT);
    const float sqrt_disc = sqrt_exact(q.discriminant);
    float root = root_div(-q.half_b - sqrt_disc, rden);
    // The far root is no smaller than the near one (the square root is >= 0 and a correctly rounded division by
    // a > 0 is monotonic), so :127-129 would reject it as well: a sphere behind the closest hit costs one root.
    bool accepted = !(root > closest);
    if (accepted && root < tmin) {
        st.section(ST_FAR_ROOT);
        root = root_div(-q.half_b + sqrt_disc, rden);
        accepted = !(root < tmin || root > closest);
    }
    bool take = accepted && root < closest;
    if (accepted && !take) {
        RT_COLD_PATH();
        take = best < 0 || slot_sphere_index(slot_spheres, slot) > slot_sphere_index(slot_spheres, best);
    }
    // (selects: as branches, the early return and the assignment cost more than the few instructions they skip)
    closest = take ? root : closest;
    best = take ? slot : best;
}

// One child of a box group against the ray: whether the parameter interval inside the box, grown by the ray's
// margin, reaches into [0, closest].  lo/hi are the child's bounds on the three axes; the arithmetic and why
// its rounding is harmless are described in rt_tree.cpp.  Returns a key for choosing the child to visit first:
// the bits of the entry parameter (>= 0, so they order like the floats) with the child's number k in the two lowest
// bits, or all ones for a child the ray does not reach.
// nx .. fz: the child's bounds on the three axes, on each axis the one the ray crosses first and the one it crosses
// last (the low and the high bound, swapped where the ray runs towards lower coordinates).  The two slab
// parameters of an axis are ordered by construction -- fma is monotonic in each operand, the low plane's offset
// is the smaller one exactly when inv >= 0 -- so no minimum or maximum per axis is needed.
__device__ __forceinline__ uint32_t box_reached(float nx, float ny, float nz, float fx, float fy, float fz, const TreeWalk& t, uint32_t k)
{
    const float ax = __builtin_fmaf(nx, t.inv.x, t.near_off.x), bx = __builtin_fmaf(fx, t.inv.x, t.far_off.x);
    const float ay = __builtin_fmaf(ny, t.inv.y, t.near_off.y), by = __builtin_fmaf(fy, t.inv.y, t.far_off.y);
    const float az = __builtin_fmaf(nz, t.inv.z, t.near_off.z), bz = __builtin_fmaf(fz, t.inv.z, t.far_off.z);
    const float tn = fmaxf(fmaxf(fmaxf(ax, ay), az), 0.0f);
    const float tf = fminf(fminf(fminf(bx, by), bz), t.closest);
    return tn <= tf ? ((__float_as_uint(tn) & ~3u) | k) : 0xFFFFFFFFu;
}

// Walk of the complete four-way tree (rt_tree.hpp: QuadTree), resumable like tree_walk.  One trip = one group
// of boxes and, for a lane that comes down to the last level in that trip, the group of four spheres there as
// well.  The children of a box group whose box the ray reaches are visited one after the other, the not yet
// visited ones waiting as four bits of `stack`; the spheres of a last-level group are fetched together and those
// whose discriminant allows a hit go through the root arithmetic one after the other.  The tree is implicit
// (child k of group g is 4 g + roots + k), so going down, across and up is arithmetic on the group index and needs
// no memory access.  `best` is a slot of the sphere list in slot order.
// tree = the box groups (six float4 each), then the groups of spheres (four float4 each), the prologue last.
template <class Stats>
__device__ __forceinline__ void quad_walk(const TraceArgs& A, const float4* tree, bool live, f3 ro, f3 rd, TreeWalk& t, Stats& st)
{
    const uint32_t roots = A.quad_roots, first_leaf = A.quad_first_leaf;
    const float4* sphere_groups = tree + 2u * first_leaf; // group g of the last level starts at sphere_groups + 4 g
    if (live && t.grp == kQuadIdle) {
        // a new segment
        t.closest = 1e20f;
        t.best = -1;
        t.rden = ray_denominator(ro, rd, true); // a scene with a hierarchy is bounded
        // the box tests of the hierarchy are proven for rays in the bounded range only
        if (t.rden.bounded) {
            // the spheres kept out of the tree first: a hit there bounds the walk from its first group
            const float4* prologue = sphere_groups + 4u * (A.tree_nodes - 1u);
            for (uint32_t k = 0; k < A.quad_prologue; ++k) {
                const SphereQuadratic q = sphere_quadratic(prologue[k], ro, rd, t.rden.a);
                if (!(q.discriminant < 0.0f) && !behind_origin(q.half_b, q.c)) {
                    sphere_closest_slot(q, t.rden, kTMin, (int)(4u * (A.tree_nodes - 1u - first_leaf) + k), A.spheres, t.closest, t.best, st);
                }
            }
            // this ray's box margin and slab constants (rt_tree.cpp)
            const KernelArgs M = args_in_memory();
            const f3 from_centre = ro - mk(M->quad_bound.x, M->quad_bound.y, M->quad_bound.z);
            const float reach = __builtin_fmaf(__builtin_amdgcn_sqrtf(dot(from_centre, from_centre)), 1.0001f, M->quad_bound.w);
            const float margin = __builtin_fmaf(reach, fminf(0.003f, __builtin_fmaf(reach, M->quad_margin, 2e-6f)),
                                                3e-7f * fmaxf(fmaxf(fabsf(ro.x), fabsf(ro.y)), fabsf(ro.z)));
            t.inv = mk(__builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(rd.x), -0x1p90f, 0x1p90f),
                       __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(rd.y), -0x1p90f, 0x1p90f),
                       __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(rd.z), -0x1p90f, 0x1p90f));
            const f3 lo_off = mk(-(ro.x + margin) * t.inv.x, -(ro.y + margin) * t.inv.y, -(ro.z + margin) * t.inv.z);
            const f3 hi_off = mk(-(ro.x - margin) * t.inv.x, -(ro.y - margin) * t.inv.y, -(ro.z - margin) * t.inv.z);
            // where the ray runs towards lower coordinates it crosses a box's high plane first (the sign bit, so that
            // a direction component of -0, whose reciprocal is clamped to -2^90, counts as negative like its inv)
            const bool neg_x = (int32_t)__float_as_uint(t.inv.x) < 0, neg_y = (int32_t)__float_as_uint(t.inv.y) < 0,
                       neg_z = (int32_t)__float_as_uint(t.inv.z) < 0;
            t.near_off = mk(neg_x ? hi_off.x : lo_off.x, neg_y ? hi_off.y : lo_off.y, neg_z ? hi_off.z : lo_off.z);
            t.far_off = mk(neg_x ? lo_off.x : hi_off.x, neg_y ? lo_off.y : hi_off.y, neg_z ? lo_off.z : hi_off.z);
            t.sel_x = neg_x ? 48u : 0u;
            t.sel_y = neg_y ? 48u : 0u;
            t.sel_z = neg_z ? 48u : 0u;
            t.stack = 0u;
            t.root = 0u;
            t.grp = 0u;
        } else {
            RT_COLD_PATH();
            const uint32_t slots = 4u * (A.tree_nodes - first_leaf);
            for (uint32_t s = 0; s < slots; ++s) {
                const float4* rec = reinterpret_cast<const float4*>(A.spheres + s);
                if (__float_as_uint(rec[1].w) == 0xFFFFFFFFu) {
                    continue; // an unused slot
                }
                const float4 cr = rec[0];
                const SphereQuadratic q = sphere_quadratic(make_float4(cr.x, cr.y, cr.z, cr.w * cr.w), ro, rd, t.rden.a);
                if (!(q.discriminant < 0.0f)) {
                    sphere_closest_slot(q, t.rden, kTMin, (int)s, A.spheres, t.closest, t.best, st);
                }
            }
        }
    }
    // the round ends when fewer than this many lanes are still walking
    const uint32_t in_loop = (uint32_t)__popcll(wave_ballot(true));
    const uint32_t few = __builtin_amdgcn_readfirstlane(kQuadWalkQuorum != 0u && in_loop > kQuadWalkQuorum ? in_loop - kQuadWalkQuorum : 0u);
    unsigned long long walking = wave_ballot(t.grp != kQuadIdle);
    if ((uint32_t)__popcll(walking) <= few) {
        return;
    }
    do {
        bool done_here = false; // nothing (more) to do below this lane's group: on to a waiting sibling or the next root
        if (t.grp < first_leaf) {
            // (an idle lane's group is 0xFFFFFFFF)
            st.section(ST_NODE);
            // a group: the low bounds of its four children on x, y, z (three float4), then the high bounds; per axis the
            // bound the ray crosses first is read at a per-lane offset of 0 or 48 bytes, the other one at the opposite
            const char* b = reinterpret_cast<const char*>(tree) + 96u * t.grp;
            const float4 nx = *reinterpret_cast<const float4*>(b + t.sel_x), fx = *reinterpret_cast<const float4*>(b + 48u - t.sel_x);
            const float4 ny = *reinterpret_cast<const float4*>(b + 16u + t.sel_y), fy = *reinterpret_cast<const float4*>(b + 64u - t.sel_y);
            const float4 nz = *reinterpret_cast<const float4*>(b + 32u + t.sel_z), fz = *reinterpret_cast<const float4*>(b + 80u - t.sel_z);
            const uint32_t k0 = box_reached(nx.x, ny.x, nz.x, fx.x, fy.x, fz.x, t, 0u);
            const uint32_t k1 = box_reached(nx.y, ny.y, nz.y, fx.y, fy.y, fz.y, t, 1u);
            const uint32_t k2 = box_reached(nx.z, ny.z, nz.z, fx.z, fy.z, fz.z, t, 2u);
            const uint32_t k3 = box_reached(nx.w, ny.w, nz.w, fx.w, fy.w, fz.w, t, 3u);
            const uint32_t pass = (k0 != 0xFFFFFFFFu ? 1u : 0u) | (k1 != 0xFFFFFFFFu ? 2u : 0u) | (k2 != 0xFFFFFFFFu ? 4u : 0u) |
                                  (k3 != 0xFFFFFFFFu ? 8u : 0u);
            if (pass != 0u) {
                // Down to the child the ray enters first (a hit found there spares the walk the children behind it);
                // the others wait, and are visited in the order of their numbers.  The order of the visits does not
                // change the result (sphere_closest_slot).
                const uint32_t first = min(min(k0, k1), min(k2, k3)) & 3u;
                t.stack = (t.stack << 4) | (pass & ~(1u << first));
                t.grp = 4u * t.grp + roots + first;
            } else {
                done_here = true;
            }
        }
        // (a lane whose box group failed keeps that group, which lies above the last level; an idle lane's is all ones)
        const bool at_spheres = t.grp - first_leaf < kQuadIdle - first_leaf;
        bool spheres_now = true;
        if (kLeafShare != 0u) {
            // The groups of spheres are tested when enough of the walking lanes stand at one; until then those lanes
            // keep their place.  (The block costs the wave the same however few lanes need it, and a trip that tests
            // both kinds of group costs more than twice the box block alone.)  When no lane has a box group to go on
            // with, every walking lane stands at a group of spheres and the quorum, a share of them, is met.
            static_assert(kLeafShare <= 8u, "the quorum must not exceed the walking lanes");
            const uint32_t waiting = (uint32_t)__popcll(wave_ballot(at_spheres));
            const uint32_t quorum = ((uint32_t)__popcll(walking) * kLeafShare) >> 3; // (the box block leaves no lane idle)
            spheres_now = waiting >= quorum;
        }
        if (at_spheres && spheres_now) {
            st.section(ST_SPHERE_TEST);
            const float4* g = sphere_groups + 4u * t.grp;
            const float4 b0 = g[0], b1 = g[1], b2 = g[2], b3 = g[3];
            const SphereQuadratic q0 = sphere_quadratic(b0, ro, rd, t.rden.a);
            const SphereQuadratic q1 = sphere_quadratic(b1, ro, rd, t.rden.a);
            const SphereQuadratic q2 = sphere_quadratic(b2, ro, rd, t.rden.a);
            const SphereQuadratic q3 = sphere_quadratic(b3, ro, rd, t.rden.a);
            // A sphere goes on to the root arithmetic unless its discriminant is negative or it lies behind the origin
            // (behind_origin: half_b and c both with a clear sign bit).  All of it is sign bits here: rays that walk the
            // tree are in the bounded range, so no discriminant is a NaN, and none is -0 either (half_b * half_b is never
            // -0, and x - y is -0 only for x = -0) -- `discriminant < 0` is the sign bit of the discriminant.
            const uint32_t t0 = (__float_as_uint(q0.half_b) | __float_as_uint(q0.c)) & ~__float_as_uint(q0.discriminant);
            const uint32_t t1 = (__float_as_uint(q1.half_b) | __float_as_uint(q1.c)) & ~__float_as_uint(q1.discriminant);
            const uint32_t t2 = (__float_as_uint(q2.half_b) | __float_as_uint(q2.c)) & ~__float_as_uint(q2.discriminant);
            const uint32_t t3 = (__float_as_uint(q3.half_b) | __float_as_uint(q3.c)) & ~__float_as_uint(q3.discriminant);
            const uint32_t pass = (t0 >> 31) | ((t1 >> 30) & 2u) | ((t2 >> 29) & 4u) | ((t3 >> 28) & 8u);
            const int slot0 = (int)(4u * (t.grp - first_leaf));
            for (uint32_t m = pass; m != 0u; m &= m - 1u) {
                const uint32_t i = (uint32_t)__builtin_ctz(m);
                // the two values the root arithmetic needs, picked by two levels of selects: as branches (which is what
                // an assignment of the whole record under `if` compiles to) the choice cost 2 % of the 4096-sphere launch
                const bool odd = (i & 1u) != 0u, upper = i >= 2u;
                const float hb_lo = odd ? q1.half_b : q0.half_b, hb_hi = odd ? q3.half_b : q2.half_b;
                const float d_lo = odd ? q1.discriminant : q0.discriminant, d_hi = odd ? q3.discriminant : q2.discriminant;
                SphereQuadratic q;
                q.half_b = upper ? hb_hi : hb_lo;
                q.discriminant = upper ? d_hi : d_lo;
                q.c = 0.0f; // not used from here on
                sphere_closest_slot(q, t.rden, kTMin, slot0 + (int)i, A.spheres, t.closest, t.best, st);
            }
            done_here = true;
        }
        if (done_here) {
            if (t.stack == 0u) {
                // nothing waits below this root: on to the next one
                t.root += 1u;
                t.grp = t.root < roots ? t.root : kQuadIdle;
            } else {
                // Up to the nearest level where a sibling waits, and across to it.  The parent of group g is
                // (g - roots) >> 2, so its ancestor `up` levels higher is (g - roots * (4^up - 1) / 3) >> (2 up) -- the
                // subtrahends of the single steps add up, each shifted by the steps that follow it, and the floors
                // compose -- with (4^up - 1) / 3 = the `up` lowest bits of 0x55555555...: no loop over the levels.
                const uint32_t c = (uint32_t)__builtin_ctz(t.stack);
                const uint32_t up4 = c & ~3u;                                     // 4 * up
                const uint32_t up2 = up4 >> 1;                                    // 2 * up
                const uint32_t at = (t.grp - roots * (0x55555555u & ((1u << up2) - 1u))) >> up2;
                t.stack >>= up4;
                t.stack &= t.stack - 1u;
                t.grp = ((at - roots) & ~3u) + roots + (c & 3u);
            }
        }
        walking = wave_ballot(t.grp != kQuadIdle);
    } while ((uint32_t)__popcll(walking) > few);
}

// A unit of work: a pixel tile x a chunk of frames, pooled by one wave.  Its description lives in
// LDS behind the wave's hash tables (it is only needed when a lane decodes a new item, and scalar
// registers are scarce in the trace loop); only the item count stays in a register.
struct Unit {
    uint32_t x0, y0;         // first pixel of the tile
    uint32_t tile_w, tile_h; // :302-305 bounds (ragged right/bottom edges)
    uint32_t tile_pixels;
    uint32_t rel_begin;      // first frame of the chunk, relative to first_frame
    uint32_t n_items;
};
constexpr uint32_t kUnitWords = 96; // offset of the unit description in a wave's table
constexpr uint32_t kWaveTableWords = 104;

// The next unit of the launch for this wave, from the counter the host zeroes before the launch.
// Every lane of the wave has to be active.
__device__ __forceinline__ uint32_t next_unit(uint32_t* counter)
{
    uint32_t unit = 0;
    if ((threadIdx.x & 63u) == 0u) {
        unit = atomicAdd(counter, 1u);
    }
    return __builtin_amdgcn_readfirstlane(unit);
}

// Unit number `unit` (chunk-major) of the launch, and the wave's tables of the hash terms of :37-44
// for its columns, rows and frames (the seed of an item is hash(px) ^ hash(py) ^ hash(frame),
// :310-311).  Every lane of the wave has to be active.
template <bool STRIPES>
__device__ __forceinline__ uint32_t load_unit(const TraceArgs& A_in_registers, uint32_t unit, uint32_t* hash_tab)
{
    (void)A_in_registers;
    const KernelArgs A = args_in_memory();
    Unit u;
    const uint32_t chunk = unit / A->n_tiles;
    const uint32_t tile = unit - chunk * A->n_tiles;
    const uint32_t tile_y = tile / A->tiles_x;
    const uint32_t tile_x = tile - tile_y * A->tiles_x;
    // the pixel tile of a unit is 2^tile_w_log2 x 2^tile_h_log2, at most 16 x 16
    u.x0 = tile_x << A->tile_w_log2;
    u.tile_w = min(1u << A->tile_w_log2, A->width - u.x0);
    const uint32_t rel_row = tile_y << A->tile_h_log2;
    if (!STRIPES) {
        u.y0 = A->row_begin + rel_row;
        u.tile_h = min(1u << A->tile_h_log2, A->row_end - u.y0);
    } else {
        // interleaved stripes: the launch's rows are numbered through its own stripes (the host keeps
        // the tile height a divisor of stripe_rows, so a tile never straddles two stripes)
        const uint32_t k = rel_row / A->stripe_rows;
        const uint32_t top = (A->stripe_first + k * A->stripe_stride) * A->stripe_rows;
        u.y0 = top + (rel_row - k * A->stripe_rows);
        const uint32_t bottom = min(top + A->stripe_rows, A->height); // the last stripe of the image may be cut short
        u.tile_h = u.y0 < bottom ? min(1u << A->tile_h_log2, bottom - u.y0) : 0u;
    }
    u.tile_pixels = u.tile_w * u.tile_h;
    u.rel_begin = chunk * A->frames_per_unit;
    const uint32_t rel_count = min(A->frames_per_unit, A->n_frames - u.rel_begin);
    u.n_items = u.tile_pixels * rel_count;
    const uint32_t lane = threadIdx.x & 63u;
    if (lane < 32u) {
        hash_tab[lane] = hash1(lane < 16u ? u.x0 + lane : u.y0 + lane - 16u);
    }
    if (lane < rel_count) {
        hash_tab[32u + lane] = hash1(A->first_frame + (u.rel_begin + lane) * A->frame_stride);
    }
    if (lane == 0u) {
        hash_tab[kUnitWords + 0u] = u.x0;
        hash_tab[kUnitWords + 1u] = u.y0;
        hash_tab[kUnitWords + 2u] = u.tile_w;
        hash_tab[kUnitWords + 3u] = u.tile_h;
        hash_tab[kUnitWords + 4u] = u.tile_pixels;
        hash_tab[kUnitWords + 5u] = u.rel_begin;
    }
    return u.n_items;
}

// BLOCK = threads per workgroup: 64 for scenes of up to 16 spheres, 256 for larger ones whose whole
// records fit in LDS, 512 for the geometry-only layout (64 KiB of LDS per workgroup: two workgroups,
// 16 waves, per CU); the hierarchy uses 256, 512 or 1024 depending on its size.
// PERSISTENT = kOneUnit: the launch has as many waves as units (for the benchmark, waves that take
// their next unit from a counter when their pool runs dry were measured 2.8 % slower);
// kByCounter: the hierarchy kernels, see below.
// STRIPES: the launch renders interleaved stripes of rows (TraceArgs::stripe_rows != 0) instead of one band.
template <int MODE, class Stats, int BLOCK, int PERSISTENT, bool STRIPES>
__global__ __launch_bounds__(BLOCK, (MODE == kTreeQuadGlobal || (MODE == kTreeQuad && BLOCK <= 512)) ? RT_QUAD_WAVES_PER_SIMD : RT_WAVES_PER_SIMD) RT_SGPR_ATTR void rt_trace_kernel(TraceArgs A)
{
    constexpr bool MAT_IN_LDS = MODE == kFullLds || MODE == kFullLds5;
    extern __shared__ float4 lds[];
    const uint32_t sphere_count = MODE == kFullLds5 ? 5u : A.params.frame_sample_depth_count[3];
    // records ahead of the hash tables: the scene (1 or 4 float4 per sphere) or the hierarchy
    // (a float4 and a word per node)
    constexpr bool QUAD = MODE == kTreeQuad || MODE == kTreeQuadGlobal;
    constexpr bool TREE = MODE == kTree || MODE == kTreeGlobal || QUAD;
    constexpr bool IN_GLOBAL = MODE == kTreeGlobal || MODE == kGeomGlobal || MODE == kTreeQuadGlobal; // nothing but the hash tables in LDS
    const uint32_t lds_records = IN_GLOBAL ? 0u : (MODE == kTree ? A.tree_nodes : (MODE == kTreeQuad ? 4u * A.tree_nodes + 2u * A.quad_first_leaf : sphere_count));
    float4* lds_geom = lds;
    float4* mat0 = lds + sphere_count;
    float4* mat1 = lds + 2 * sphere_count;
    float4* hit_rec = lds + 3 * sphere_count; // kFullLds only
    uint32_t* lds_word = reinterpret_cast<uint32_t*>(lds + lds_records);
    const float4* geom = MODE == kTreeGlobal || MODE == kTreeQuadGlobal ? A.tree_geom : (MODE == kGeomGlobal ? A.loop_geom : lds_geom);
    const uint32_t* node_word = MODE == kTreeGlobal ? A.tree_word : lds_word;
    // per-wave tables of the three hash terms of :37-44 for this unit's columns, rows and frames
    // (the seed of an item is hash(px) ^ hash(py) ^ hash(frame), :310-311)
    uint32_t* hash_tab = (MODE == kTree ? lds_word + lds_records
                          : MODE == kTreeQuad ? lds_word
                                        : reinterpret_cast<uint32_t*>(lds + (IN_GLOBAL ? 0u : (MAT_IN_LDS ? 4u : 1u)) * sphere_count)) +
                         (threadIdx.x >> 6) * kWaveTableWords;

    Stats st;

    // ---- this wave's (first) unit ----
    // With the hierarchy a wave is persistent: it takes one unit after another from a counter in memory
    // and moves on to the next as soon as its pool is empty, while some of its lanes are still busy with
    // items of the old one (walks are long and uneven, and a workgroup of sixteen waves holds its LDS
    // until its last wave is through).  Otherwise there are as many waves as units: for the small scenes
    // persistent waves were measured slower, with the counter (one atomic per unit: 0.06 -> 0.29 ms for
    // a 720p frame) as well as with unit numbers a launch-width apart (1080p frame 0.88 -> 1.08 ms).
    const uint32_t n_units = A.n_tiles * A.n_chunks;
    const uint32_t wave_index = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    uint32_t unit = PERSISTENT == kByCounter ? next_unit(A.unit_counter) : wave_index;
    const uint32_t first_unit = unit;
    const bool spare_wave = first_unit >= n_units;
    const uint32_t full_w = 1u << A.tile_w_log2, full_h = 1u << A.tile_h_log2;
    uint32_t n_items = load_unit<STRIPES>(A, min(first_unit, n_units - 1u), hash_tab);
    uint32_t next_item = 0; // wave-uniform pool cursor
    bool more_units = PERSISTENT != kOneUnit; // wave-uniform: the last unit has not been passed yet
    // both end with the workgroup barrier
    if (MODE == kTree) {
        stage_tree(A.tree_geom, A.tree_word, A.tree_nodes, lds_geom, lds_word);
    } else if (MODE == kTreeQuad) {
        stage_quad_tree(A.tree_geom, lds_records, lds_geom);
    } else if (!IN_GLOBAL) {
        stage_scene<MAT_IN_LDS>(A.spheres, sphere_count, lds_geom, mat0, mat1, hit_rec);
    }
    if (spare_wave) {
        return; // the last workgroup may have more waves than units are left
    }

    const uint32_t spp_raw = A.params.frame_sample_depth_count[1];
    const uint32_t spp = spp_raw > 1u ? spp_raw : 1u; // :308
    const uint32_t max_depth = A.params.frame_sample_depth_count[2];

    const f3 origin = mk(A.params.origin_lens[0], A.params.origin_lens[1], A.params.origin_lens[2]);
    const float lens_radius = A.params.origin_lens[3];
    const f3 lower_left = mk(A.params.lower_left[0], A.params.lower_left[1], A.params.lower_left[2]);
    const f3 horizontal = mk(A.params.horizontal[0], A.params.horizontal[1], A.params.horizontal[2]);
    const f3 vertical = mk(A.params.vertical[0], A.params.vertical[1], A.params.vertical[2]);
    const f3 cam_u = mk(A.params.u[0], A.params.u[1], A.params.u[2]);
    const f3 cam_v = mk(A.params.v[0], A.params.v[1], A.params.v[2]);
    const float inv_w = A.params.inv_resolution[0];
    const float inv_h = A.params.inv_resolution[1];
    const float res_y_minus_1 = A.params.resolution[1] - 1.0f; // float(params.resolution.y - 1u ...), :320
    // With a lens radius of zero the lens offset of :324-325 is +-0 in every component.  Subtracting
    // it from the direction and adding it to the origin then changes nothing, except the sign of a
    // zero: origin + (+-0) can differ from origin only where an origin component is itself zero, and
    // (x - origin) - (+-0) only where x - origin is -0, which needs a zero origin component too
    // (an exact cancellation rounds to +0).  So for a pinhole camera whose position has no zero
    // component both steps are skipped; the two draws of randomInUnitDisk are still consumed.
    const bool pinhole = lens_radius == 0.0f && origin.x != 0.0f && origin.y != 0.0f && origin.z != 0.0f;

    // Small scenes: which spheres a camera ray of this unit's tile can hit at all.  The rays of the tile
    // leave the camera position (or a point of the lens, at most 1.5 lens radii from it) in directions
    // dc + delta, |delta| <= rho, dc the direction through the middle of the tile.  A ray at angle phi to
    // w = centre - origin passes the centre at distance |w| sin(phi), and sin is 1-Lipschitz, so every
    // ray of the tile stays at least q_c - |w| theta away, q_c the distance of the middle ray and
    // theta <= 1.6 rho / |dc| the half angle of the bundle.  hitSphere's discriminant can only come out
    // non-negative when that distance is below r + 0.0014 |w| (ten roundings of relative 2^-24 in
    // :112-117 for rays of unit length and a bounded scene); with 0.002 |w| and another 0.1 % of margin
    // the rounding of this estimate itself (relative 1e-6) does not matter.  A NaN anywhere keeps the sphere.
    // (rho < |dc| / 4 is required of both tests.)
    constexpr bool CULL = MAT_IN_LDS && BLOCK == 64;
    const uint32_t full_mask = CULL ? (1u << sphere_count) - 1u : 0xFFFFFFFFu; // at most 16 spheres
    uint32_t camera_mask = full_mask;
    if (CULL && A.scene_bounded != 0u && (fabsf(origin.x) + fabsf(origin.y)) + fabsf(origin.z) < 0x1p30f) {
        const uint32_t lane = threadIdx.x & 63u;
        bool keep = lane < sphere_count;
        if (lane < sphere_count) {
            const float tile_x0 = (float)hash_tab[kUnitWords + 0u], tile_y0 = (float)hash_tab[kUnitWords + 1u];
            const float half_w = 0.5f * (float)hash_tab[kUnitWords + 2u], half_h = 0.5f * (float)hash_tab[kUnitWords + 3u];
            const float uc = (tile_x0 + half_w) * inv_w;
            const float vc = ((A.params.resolution[1] - tile_y0) - half_h) * inv_h; // :320 flips y
            const f3 dc = lower_left + horizontal * uc + vertical * vc - origin;
            const float lens = 1.5f * fabsf(lens_radius);
            const float rho = ((half_w * inv_w) * sqrtf(dot(horizontal, horizontal)) + (half_h * inv_h) * sqrtf(dot(vertical, vertical)) +
                               lens) * 1.001f;
            const float4 g = geom[lane];
            const f3 w = mk(g.x, g.y, g.z) - origin;
            const f3 cr = mk(w.y * dc.z - w.z * dc.y, w.z * dc.x - w.x * dc.z, w.x * dc.y - w.y * dc.x);
            const float len_dc = sqrtf(dot(dc, dc)), len_w = sqrtf(dot(w, w));
            const float q_c = sqrtf(dot(cr, cr)) / len_dc;
            const float reach = (sqrtf(g.w) + lens) * 1.001f;
            const bool misses = rho < 0.25f * len_dc && q_c - len_w * (1.6f * rho / len_dc) - 0.002f * len_w > reach;
            // Or the sphere lies behind every ray: origin outside it and dot(oc, d) > 0 by a margin, so
            // that :121-131 find both roots negative (the lines of such rays do cross the ground sphere).
            const bool behind = rho < 0.25f * len_dc && len_w - lens > 1.01f * sqrtf(g.w) &&
                                dot(w, dc) + len_w * rho + (0.002f * len_w + lens) * len_dc < 0.0f;
            keep = !(misses || behind);
        }
        camera_mask = (uint32_t)wave_ballot(keep);
    }

    // ---- per-lane state ----
    bool has_item = false;
    bool live = false;         // the lane carries a ray that has to be traced in this iteration
    bool pending_norm = false; // rd still has to be normalized (camera ray :328, lambert :159)
    uint32_t samples_left = 0;
    uint32_t rel = 0;          // frame of the current item, relative to first_frame
    uint32_t rng = 0;
    uint32_t depth = 0;
    uint32_t texel = 0;        // py * width + px (the host keeps width * height below 2^32)
    float fpx = 0.0f, fpy_flipped = 0.0f;
    f3 frame_sum = mk(0.0f, 0.0f, 0.0f); // `accum` of :313
    f3 ro = origin, rd = origin;
    f3 throughput = mk(1.0f, 1.0f, 1.0f);

    TreeWalk walk; // hierarchy only
    walk.at = 0xFFFFFFFFu;
    walk.g = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    walk.w = 0u;
    walk.link = 0u;
    walk.closest = 1e20f;
    walk.best = -1;
    walk.beyond_limit = 0.0f;
    walk.rden.a = 1.0f;
    walk.rden.rcp_a = 1.0f;
    walk.rden.a_ok = true;
    walk.rden.bounded = false;
    walk.grp = kQuadIdle;
    walk.stack = 0u;
    walk.root = 0u;
    walk.inv = walk.near_off = walk.far_off = mk(0.0f, 0.0f, 0.0f);
    walk.sel_x = walk.sel_y = walk.sel_z = 0u;

    for (;;) {
        st.section(ST_ITER);
        st.phase_start();
        // ------------------------------------------------------------------ merged normalize #2
        if (pending_norm) {
            st.section(ST_NORM2);
            rd = normalize(rd); // :328 (camera) or :159 (lambert)
            pending_norm = false;
        }

        st.phase_end(PH_NORM2);
        // ------------------------------------------------------------------ closest hit, :244-259
        float closest = 1e20f;
        int best = -1;
        RayDenominator rden;
        rden.a = 1.0f;
        rden.rcp_a = 1.0f;
        rden.a_ok = true;
        rden.bounded = false;
        bool traced = live; // the lane's segment has been searched to its end in this iteration
        if (Stats::kEnabled && live && wave_ballot(live && depth != 0u) == 0ull) {
            st.section(ST_ALL_PRIMARY); // every ray of the wave in this iteration is a camera ray
        }
        if (QUAD) {
            quad_walk(A, geom, live, ro, rd, walk, st);
            traced = live && walk.grp == kQuadIdle;
            closest = walk.closest;
            best = traced ? walk.best : -1;
            rden = walk.rden;
        } else if (TREE) {
            tree_walk<MODE == kTreeGlobal>(A, geom, node_word, sphere_count, live, ro, rd, walk, st);
            traced = live && walk.at >= A.tree_nodes * 16u;
            closest = walk.closest;
            best = traced ? walk.best : -1;
            rden = walk.rden;
        } else if (live) {
            // when every ray of the wave is a camera ray of this unit, only the spheres its tile can see
            uint32_t sphere_mask = 0xFFFFFFFFu;
            if (CULL) {
                // (the loop over a mask is 1.4 % slower than the counted loop when it has to visit every
                // sphere anyway, so it is only used with a reduced mask)
                sphere_mask = wave_ballot(live && depth != 0u) == 0ull && camera_mask != full_mask ? camera_mask : 0xFFFFFFFFu;
                if (MODE == kFullLds5) {
                    sphere_mask &= 31u; // one loop for both cases there
                }
            }
            closest_hit<MODE>(A, geom, node_word, sphere_count, sphere_mask, ro, rd, rden, closest, best, st);
        }
        st.phase_end(PH_INTERSECT);
        const bool hit = best >= 0; // only lanes whose segment has been searched can hit
        const bool miss = traced && !hit;

        // ------------------------------------------------------------------ winner's record, :135-149
        f3 p, n, albedo; // defined for hit lanes only
        bool front = false;
        uint32_t mat = 3u;
        float fuzz, ref_idx, flags;
        if (hit) {
            st.section(ST_HIT);
            float4 g, m0, m1;
            if (MAT_IN_LDS) {
                g = hit_rec[best];
                m0 = mat0[best];
                m1 = mat1[best];
            } else {
                const float4* src = reinterpret_cast<const float4*>(A.spheres + best);
                const float4 cr = src[0];
                const float4 al = src[1];
                g = cr; // only the centre is used
                m0 = make_float4(al.x, al.y, al.z, cr.w);
                m1 = src[2];
            }
            if (MAT_IN_LDS) {
                hit_finish_with(mk(g.x, g.y, g.z), m0.w, g.w, ro, rd, closest, p, n, front);
            } else {
                hit_finish(mk(g.x, g.y, g.z), m0.w, ro, rd, closest, p, n, front);
            }
            mat = MAT_IN_LDS ? __float_as_uint(m1.x) : (uint32_t)(m1.x + 0.5f); // :145
            albedo = mk(m0.x, m0.y, m0.z);
            fuzz = m1.y;
            ref_idx = m1.z;
            flags = m1.w;
        }
        const bool is_lambert = hit && mat == 0u;
        const bool is_metal = hit && mat == 1u;
        st.phase_end(PH_HIT);

        // ------------------------------------------------------------------ item bookkeeping
        // Lanes whose path is over (it ended earlier, or it just left the scene) start their next
        // camera ray in this iteration.  If the item has no samples left it is complete, apart from
        // the sky term a missing lane adds below: fetch the next item now (its random stream has to
        // exist before the samplers run) and write the finished one out after the shading.
        // Lanes whose path is over start a new camera sample only when at least kRegenQuorum of them
        // want to, or when no lane of the wave is in the middle of a path: the blocks that only such
        // lanes run (item fetch, jitter, camera ray) then execute for many lanes at a time or not at
        // all.  Measured on the benchmark: no quorum 59.2 Gsamples/s, 8: 59.2, 12: 60.1, 16: 60.5,
        // 20: 60.3, 24: 59.9, 32: 59.3, 48: 57.2.
        const bool path_done = !live || miss;
        bool regen = path_done;
        // (not with the hierarchy: there a waiting lane idles through a walk of a hundred nodes, which
        // costs more than the blocks it would share are worth -- measured 65.3 against 62.4 ms on one box)
        if (kRegenQuorum != 0u && !TREE) {
            const unsigned long long done_mask = wave_ballot(path_done);
            const unsigned long long in_loop = wave_ballot(true);
            regen = path_done && ((uint32_t)__popcll(done_mask) >= kRegenQuorum || done_mask == in_loop);
        }
        const bool need_item = regen && samples_left == 0u;
        const bool write_out = need_item && has_item;
        const uint32_t old_texel = texel;
        const uint32_t old_rel = rel;
        if (write_out) {
            const KernelArgs M = args_in_memory();
            if (M->rng_out != nullptr && rel + 1u == M->n_frames) {
                M->rng_out[texel] = rng;
            }
        }
        // Every lane still in the loop runs the pool arithmetic, so the cursor stays wave-uniform.
        const unsigned long long want = wave_ballot(need_item);
        if (PERSISTENT && want != 0ull && next_item >= n_items && more_units) {
            // The pool is empty: on to the next unit.  No lane has left the loop yet (lanes leave only
            // when there are no more units), and the lanes still busy with items of the old unit hold
            // everything they need in registers.
            unit = next_unit(args_in_memory()->unit_counter);
            if (unit < n_units) {
                n_items = load_unit<STRIPES>(A, unit, hash_tab);
                next_item = 0;
            } else {
                more_units = false;
            }
        }
        const uint32_t my_item = next_item + lane_prefix(want);
        next_item += (uint32_t)__popcll(want);
        const bool no_item = need_item && my_item >= n_items; // pool drained
        // leave after the write-out, unless there is another unit to come: then ask again next time
        const bool exiting = no_item && !more_units;
        if (PERSISTENT && no_item) {
            has_item = false;
        }
        if (need_item && !no_item) {
            st.section(ST_ITEM);
            const uint32_t unit_x0 = hash_tab[kUnitWords + 0u], unit_y0 = hash_tab[kUnitWords + 1u];
            const uint32_t tile_w = hash_tab[kUnitWords + 2u];
            const uint32_t tile_pixels = hash_tab[kUnitWords + 4u], rel_begin = hash_tab[kUnitWords + 5u];
            uint32_t frame_in_chunk, pix;
            if (tile_pixels == full_w * full_h) {
                frame_in_chunk = my_item >> (A.tile_w_log2 + A.tile_h_log2);
                pix = my_item & (tile_pixels - 1u);
            } else {
                frame_in_chunk = my_item / tile_pixels;
                pix = my_item - frame_in_chunk * tile_pixels;
            }
            uint32_t ly, lx;
            if (tile_w == full_w) {
                ly = pix >> A.tile_w_log2;
                lx = pix & (full_w - 1u);
            } else {
                ly = pix / tile_w;
                lx = pix - ly * tile_w;
            }
            const uint32_t px = unit_x0 + lx;
            const uint32_t py = unit_y0 + ly;
            rel = rel_begin + frame_in_chunk;
            texel = py * args_in_memory()->width + px;
            fpx = (float)(int32_t)px;
            fpy_flipped = res_y_minus_1 - (float)(int32_t)py;
            rng = hash_tab[lx] ^ hash_tab[16u + ly] ^ hash_tab[32u + frame_in_chunk]; // :310-311
            samples_left = spp;
            has_item = true;
        }
        const bool gen = regen && !no_item; // this lane draws a camera sample now

        // ------------------------------------------------------------------ pixel jitter, :317-320
        float u_coord = 0.0f, v_coord = 0.0f;
        if (gen) {
            st.section(ST_GEN);
            --samples_left;
            u_coord = rnd_plus(rng, fpx) * inv_w;
            v_coord = rnd_plus(rng, fpy_flipped) * inv_h;
        }

        st.phase_end(PH_ITEM);
        // ------------------------------------------------------------------ shared rejection sampler
        // randomInUnitDisk (:75-86) for the lanes that start a camera ray, randomInUnitSphere (:57-68)
        // for lambert and metal lanes, in one loop.  A draw r = k / 2^24 gives 2r - 1 = (k - 2^23) / 2^23
        // exactly, so the candidate is kept as the integer-valued floats m = k - 2^23 and the test
        // dot(p,p) >= 1 becomes (mx*mx + my*my) + mz*mz >= 2^46 (scaling by a power of two commutes
        // with every rounding involved).  Disk lanes take no third draw: their multiplier/increment
        // are (1, 0) and their mz is 0, which adds +0 to the two-term sum of :83.
        const bool want3 = is_lambert || is_metal;
        float mx = 0.0f, my = 0.0f, mz = 0.0f, m_dot = 0.0f;
        if (gen || want3) {
            st.section(ST_RIS);
            // Per-lane third draw without selects: all-ones for the lambert/metal lanes (mat 0 or 1;
            // dielectric is 2 and lanes without a hit carry 3), zero for the camera lanes.
            const uint32_t is3 = (uint32_t)((int32_t)(mat - 2u) >> 31);
            const uint32_t mul3 = (is3 & (1664525u - 1u)) + 1u;
            const uint32_t add3 = is3 & 1013904223u;
            const uint32_t mask3 = is3 & 0x00FFFFFFu;
            const int32_t off3 = (int32_t)(is3 & 0x00800000u);
            if (wave_ballot(want3) == 0ull) {
                // only camera lanes in this wave (a tile of sky, typically): two draws per candidate
                do {
                    st.section(ST_DISK_IT);
                    const uint32_t s1 = lcg_step(rng, 1664525u, 1013904223u);
                    rng = lcg_step(s1, 1664525u, 1013904223u);
                    mx = (float)((int32_t)(s1 & 0x00FFFFFFu) - 0x00800000);
                    my = (float)((int32_t)(rng & 0x00FFFFFFu) - 0x00800000);
                    m_dot = mx * mx + my * my;
                } while (m_dot >= 0x1p46f);
            } else {
                do {
                    st.section(ST_RIS_IT);
                    const uint32_t s1 = lcg_step(rng, 1664525u, 1013904223u);
                    const uint32_t s2 = lcg_step(s1, 1664525u, 1013904223u);
                    rng = lcg_step(s2, mul3, add3);
                    mx = (float)((int32_t)(s1 & 0x00FFFFFFu) - 0x00800000);
                    my = (float)((int32_t)(s2 & 0x00FFFFFFu) - 0x00800000);
                    mz = (float)((int32_t)(rng & mask3) - off3);
                    m_dot = (mx * mx + my * my) + mz * mz;
                } while (m_dot >= 0x1p46f);
            }
        }
        const f3 ris = mk(mx * 0x1p-23f, my * 0x1p-23f, mz * 0x1p-23f);
        st.phase_end(PH_SAMPLER);

        // ------------------------------------------------------------------ merged normalize #1
        // lambert: randomUnitVector = normalize(ris), :72, whose dot(ris,ris) is the sampler's last sum.
        // everything else: normalize(ray.d), :201 (metal), :215 (dielectric), :263 (sky), whose
        // dot(d,d) is the `a` of the intersection.
        const f3 unit = (is_lambert ? ris : rd) * inversesqrt(is_lambert ? m_dot * 0x1p-46f : rden.a);

        st.phase_end(PH_NORM1);
        bool scattered = true;
        if (miss) {
            // sky, :261-269
            st.section(ST_SKY);
            // Three exact rewrites with fused multiply-adds (one rounding of the same real number each):
            // 0.5 * (y + 1) = fma(0.5, y, 0.5) because halving commutes with the rounding of y + 1;
            // 0.5 * t is exact; and 0 + throughput * sky = fma(throughput, sky, +0).
            float t = __builtin_fmaf(0.5f, unit.y, 0.5f);
            float omt = 1.0f - t;
            f3 sky = mk(__builtin_fmaf(0.5f, t, omt), omt + 0.7f * t, omt + 1.0f * t);
            f3 radiance = mk(__builtin_fmaf(throughput.x, sky.x, 0.0f), __builtin_fmaf(throughput.y, sky.y, 0.0f),
                             __builtin_fmaf(throughput.z, sky.z, 0.0f));
            frame_sum = frame_sum + radiance; // :330
        } else if (want3) {
            if (is_lambert) {
                // scatterLambert, :155-174
                st.section(ST_LAMBERT);
                f3 target = p + n + unit;
                rd = target - p;
                pending_norm = true; // normalize(target - p) at the top of the next iteration
                if (flags > 0.5f) {
                    st.section(ST_CHECKER);
                    float checker_index = floorf(p.x) + floorf(p.z);
                    // mod(x, 2) = x - 2 * floor(x / 2); halving and doubling are exact, so this rounds once like the original
                    float checker = __builtin_fmaf(-2.0f, floorf(checker_index * 0.5f), checker_index);
                    float tone = (checker < 1.0f) ? 0.05f : 0.95f;
                    albedo = albedo * tone;
                }
            } else {
                // scatterMetal, :199-208
                st.section(ST_METAL);
                f3 reflected = reflect(unit, n);
                rd = reflected + ris * fuzz;
                scattered = dot(rd, n) > 0.0f;
            }
            // common to both: origin p + 0.001 n (:160, :203) and attenuation = albedo (:170, :206)
#ifdef RT_PK_SHADE
            ro = pk_scale_add(p, n, 0.001f);
            throughput = pk_mul(throughput, albedo);
#else
            ro = p + n * 0.001f;
            throughput = throughput * albedo;
#endif
        } else if (hit) {
            // scatterDielectric, :210-234
            st.section(ST_DIEL);
            throughput = throughput * mk(1.0f, 1.0f, 1.0f);
            float eta_ratio = front ? (MAT_IN_LDS ? fuzz : 1.0f / ref_idx) : ref_idx;
            float cos_theta = glsl_min(dot(-unit, n), 1.0f);
            float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
            bool do_reflect = eta_ratio * sin_theta > 1.0f;
            if (!do_reflect) {
                // `||` short-circuits: the draw happens only on this side
                st.section(ST_SCHLICK);
                float sch = schlick(cos_theta, eta_ratio);
                do_reflect = sch > rnd(rng);
            }
            f3 direction;
            if (do_reflect || !refract_vec(unit, n, eta_ratio, direction)) {
                direction = reflect(unit, n);
            }
            ro = p + direction * 0.001f;
            rd = direction;
        }
        if (hit) {
            ++depth;
            if (!scattered || depth >= max_depth) {
                live = false; // :287-290 break, or the bounce budget of :242 is spent
            }
        }

        st.phase_end(PH_SHADE);
        // ------------------------------------------------------------------ finished item, :333-335
        if (write_out) {
            const KernelArgs M = args_in_memory();
            if (M->partial != nullptr) {
                M->partial[(size_t)old_rel * ((size_t)M->width * M->height) + old_texel] = make_float4(frame_sum.x, frame_sum.y, frame_sum.z, 0.0f);
            } else {
                const float4 prev = M->accum[old_texel];
                const f3 total = mk(prev.x, prev.y, prev.z) + frame_sum;
                M->accum[old_texel] = make_float4(total.x, total.y, total.z, 1.0f);
                if (M->output != nullptr) {
                    // :337 (frame + 1u) * spp wraps in uint32 before the conversion
                    const uint32_t frame = M->first_frame + old_rel * M->frame_stride;
                    M->output[old_texel] = tonemap_rgba8(total, (float)((frame + 1u) * M->params.frame_sample_depth_count[1]));
                }
            }
            frame_sum = mk(0.0f, 0.0f, 0.0f);
        }
        if (exiting) {
            break;
        }
        if (miss) {
            live = false; // the path is over even if its lane has to wait for its next camera sample
        }

        // ------------------------------------------------------------------ camera ray, :322-328
        if (gen) {
            if (pinhole) {
                rd = lower_left + horizontal * u_coord + vertical * v_coord - origin;
                ro = origin;
            } else {
                float rdx = lens_radius * ris.x;
                float rdy = lens_radius * ris.y;
                f3 offset = cam_u * rdx + cam_v * rdy;
                rd = lower_left + horizontal * u_coord + vertical * v_coord - origin - offset;
                ro = origin + offset;
            }
            pending_norm = true;
            throughput = mk(1.0f, 1.0f, 1.0f);
            depth = 0u;
            live = true;
        }
        st.phase_end(PH_TAIL);
    }
    st.flush(A.stats);
}

// Adds the per-frame sums of a multi-frame launch to the accumulator in frame order (:333-335 of
// consecutive dispatches) and, if asked, runs the display tail of the last frame (:337-342).
__global__ __launch_bounds__(kBlockThreads) void rt_fold_kernel(float4* __restrict__ accum, const float4* __restrict__ partial,
                                                                uint32_t* __restrict__ output, size_t image_pixels,
                                                                uint32_t width, uint32_t height, RowSet rows, uint32_t count,
                                                                uint32_t n_frames, float sample_count)
{
    uint32_t i = blockIdx.x * kBlockThreads + threadIdx.x;
    if (i >= count) {
        return;
    }
    size_t texel;
    if (rows.stripe_rows == 0u) {
        texel = (size_t)rows.row_begin * width + i;
    } else {
        // i counts the texels of this launch's stripes, stripe after stripe
        const uint32_t stripe_texels = rows.stripe_rows * width;
        const uint32_t k = i / stripe_texels;
        const uint32_t within = i - k * stripe_texels;
        texel = (size_t)(rows.stripe_first + k * rows.stripe_stride) * stripe_texels + within;
        if (texel >= (size_t)width * height) {
            return; // the part of the last stripe below the image
        }
    }
    const float4 prev = accum[texel];
    f3 total = mk(prev.x, prev.y, prev.z);
    for (uint32_t f = 0; f < n_frames; ++f) {
        const float4 s = partial[(size_t)f * image_pixels + texel];
        total = total + mk(s.x, s.y, s.z);
    }
    accum[texel] = make_float4(total.x, total.y, total.z, 1.0f);
    if (output != nullptr) {
        output[texel] = tonemap_rgba8(total, sample_count);
    }
}

__global__ __launch_bounds__(kBlockThreads) void rt_resolve_kernel(const float4* __restrict__ accum, uint32_t* __restrict__ output,
                                                                   uint32_t first, uint32_t count, float sample_count)
{
    uint32_t i = blockIdx.x * kBlockThreads + threadIdx.x;
    if (i >= count) {
        return;
    }
    float4 v = accum[first + i];
    output[first + i] = tonemap_rgba8(mk(v.x, v.y, v.z), sample_count);
}

__global__ void rt_probe_hit_kernel(const rt_gpu_sphere* spheres, const float* rays, uint32_t n, float* out, int32_t* front_out)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    const float* r = rays + (size_t)i * 8;
    f3 o = mk(r[0], r[1], r[2]);
    f3 d = mk(r[4], r[5], r[6]);
    float tmin = r[3], tmax = r[7];
    const rt_gpu_sphere s = spheres[i];
    float radius = s.center_radius[3];
    float4 geom = make_float4(s.center_radius[0], s.center_radius[1], s.center_radius[2], radius * radius);
    float root = 0.0f;
    float* o8 = out + (size_t)i * 8;
    NoStats nost;
    bool hit = sphere_root(geom, o, d, ray_denominator(o, d, sphere_bounded(*reinterpret_cast<const float4*>(s.center_radius)) && tmin >= 1e-6f),
                           tmin, tmax, root, nost);
    f3 p = mk(0, 0, 0), nn = mk(0, 0, 0);
    bool front = false;
    if (hit) {
        hit_finish(mk(geom.x, geom.y, geom.z), radius, o, d, root, p, nn, front);
    }
    o8[0] = hit ? 1.0f : 0.0f;
    o8[1] = root;
    o8[2] = p.x; o8[3] = p.y; o8[4] = p.z;
    o8[5] = nn.x; o8[6] = nn.y; o8[7] = nn.z;
    front_out[i] = front ? 1 : 0;
}

__global__ void rt_probe_pow_kernel(const float* x, const float* y, uint32_t n, float* out)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        out[i] = glsl_pow(x[i], y[i]);
    }
}

// out[0..n) = the cheap-range helper, out[n..2n) = the full IEEE sequence, for op:
// 0: a / b (div_shared)   1: sqrt(a)   2: inversesqrt(a) = 1 / sqrt(a)
__global__ void rt_probe_arith_kernel(uint32_t op, const float* a, const float* b, uint32_t n, float* out)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    const float x = a[i], y = b[i];
    float cheap, full;
    if (op == 0u) {
        cheap = div_shared(x, y, refined_rcp(y), den_in_range(y));
        full = x / y;
    } else if (op == 1u) {
        cheap = sqrt_exact(x);
        full = sqrtf(x);
    } else {
        cheap = inversesqrt(x);
        full = 1.0f / sqrtf(x);
    }
    out[i] = cheap;
    out[n + i] = full;
}

__global__ void rt_probe_rng_kernel(const uint32_t* xyz, uint32_t n, uint32_t draws, uint32_t* out)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    uint32_t s = hash3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    uint32_t* o = out + (size_t)i * (1 + draws);
    o[0] = s;
    for (uint32_t k = 0; k < draws; ++k) {
        o[1 + k] = lcg(s);
    }
}

} // namespace

// ---- scratch of the allocation-free entry point (rt_render_frames_static) ----------------------
// Device memory that exists from the moment the code object is loaded, so that callers which must not
// call hipMalloc (under HIP graph capture, in tight serving loops) can still render several frames per
// launch.  In host code the array's name only stands for the symbol: the device address comes from
// hipGetSymbolAddress, per device, and is what kernels, memsets and copies are given.
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

namespace {
constexpr int kMaxDevices = 64;
void* g_scratch_ptr[kMaxDevices] = {};
std::mutex g_scratch_mutex;
} // namespace

hipError_t static_scratch(int device, void** ptr, size_t* bytes)
{
    if (device < 0 || device >= kMaxDevices || ptr == nullptr) {
        return hipErrorInvalidValue;
    }
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[device] == nullptr) {
        int current = -1;
        hipError_t e = hipGetDevice(&current);
        if (e != hipSuccess) {
            return e;
        }
        if (current != device) {
            return hipErrorInvalidDevice; // each device holds its own copy at its own address
        }
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) {
            return e;
        }
        g_scratch_ptr[device] = p;
    }
    *ptr = g_scratch_ptr[device];
    if (bytes != nullptr) {
        *bytes = kStaticScratchBytes;
    }
    return hipSuccess;
}

size_t trace_lds_bytes(uint32_t sphere_count, bool* mat_in_lds)
{
    // Up to 1024 spheres the whole 48-byte record lives in LDS (48 KiB, three
    // workgroups per CU); beyond that only the 16-byte geometry record does.
    const bool full = sphere_count <= kMaxSpheresFullLds;
    if (mat_in_lds != nullptr) {
        *mat_in_lds = full;
    }
    // scene records + the hash tables of up to 8 waves
    return (size_t)sphere_count * (full ? 64u : 16u) + 8u * kWaveTableWords * 4u;
}

namespace {

// Workgroups of `waves_per_block` waves and `lds` bytes that the current device holds at a time
// (160 KiB of LDS and 32 waves per compute unit).
uint64_t resident_blocks(size_t lds, uint32_t waves_per_block)
{
    // (the devices of one node are of one kind: the figure is looked up once per process)
    static std::atomic<int> cached_cus{ 0 };
    int cus = cached_cus.load();
    if (cus == 0) {
        int device = 0;
        if (hipGetDevice(&device) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) {
            (void)hipGetLastError();
            cus = 256;
        }
        cached_cus.store(cus);
    }
    const uint64_t by_lds = std::max<uint64_t>(1u, (160u * 1024u) / std::max<size_t>(lds, 1u));
    const uint64_t by_waves = std::max<uint64_t>(1u, 32u / waves_per_block);
    return static_cast<uint64_t>(cus) * std::min(by_lds, by_waves);
}

} // namespace

uint64_t resident_wave_slots()
{
    return resident_blocks(0, 1);
}

namespace {

template <int MODE, int BLOCK, int PERSISTENT>
void launch_trace_as(const TraceArgs& args, uint64_t units, size_t lds, hipStream_t stream)
{
#ifdef RT_STATS
    typedef WaveStats StatsPolicy;
#else
    typedef NoStats StatsPolicy;
#endif
    const uint32_t waves_per_block = BLOCK / 64;
    if (const char* e = std::getenv("RT_LDS_PAD")) {
        lds += (size_t)std::max(0, std::atoi(e)); // experiment: caps the resident workgroups per CU
    }
    uint64_t blocks = (units + waves_per_block - 1) / waves_per_block;
    if (PERSISTENT != kOneUnit) {
        // no more workgroups than the device holds at a time
        blocks = std::min<uint64_t>(blocks, resident_blocks(lds, waves_per_block));
    }
    if (std::getenv("RT_DEBUG_OCCUPANCY") != nullptr) {
        int per_cu = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rt_trace_kernel<MODE, StatsPolicy, BLOCK, PERSISTENT, false>, BLOCK, lds);
        fprintf(stderr, "occupancy: %d workgroups of %d threads per CU, lds %zu, blocks %llu\n", per_cu, BLOCK, lds, (unsigned long long)blocks);
    }
    if (args.stripe_rows != 0u) {
        hipLaunchKernelGGL((rt_trace_kernel<MODE, StatsPolicy, BLOCK, PERSISTENT, true>), dim3((uint32_t)blocks), dim3(BLOCK), lds, stream, args);
    } else {
        hipLaunchKernelGGL((rt_trace_kernel<MODE, StatsPolicy, BLOCK, PERSISTENT, false>), dim3((uint32_t)blocks), dim3(BLOCK), lds, stream, args);
    }
}

} // namespace

hipError_t launch_trace(const TraceArgs& args_in, hipStream_t stream)
{
    TraceArgs args = args_in;
    if (args.stripe_rows != 0u && (args.stripe_stride == 0u || (uint64_t)args.n_stripes * args.stripe_rows > 0x7FFFFFFFull)) {
        return hipErrorInvalidValue;
    }
    const uint32_t rows = args.stripe_rows != 0u ? args.n_stripes * args.stripe_rows : args.row_end - args.row_begin;
    if (rows == 0 || args.n_frames == 0) {
        return hipSuccess;
    }
    if (args.frames_per_unit == 0) {
        args.frames_per_unit = 1;
    }
    args.frames_per_unit = std::min(std::min(args.frames_per_unit, 64u), args.n_frames);
    if (args.tile_w_log2 > 4u || args.tile_h_log2 > 4u) {
        // Not chosen by the caller: a unit pools about 512 items, from as many frames and as few
        // pixels as the launch allows (between 2 x 4 and 8 x 8 pixels; 8 x 16 where the host asks for it).  Measured on the benchmark
        // (64 frames): 8 x 8 pixels x 16 frames 55.0 Gsamples/s, 8 x 8 x 8 56.6, 4 x 4 x 16 57.2,
        // 2 x 4 x 32 57.8, 2 x 4 x 64 58.2; the number of VALU instructions is the same in all of
        // them, the shorter and more numerous units keep more waves resident.
        // With the hierarchy units are handed out by a counter, so they can be much smaller: about 32
        // items.  Rays of neighbouring pixels, and of one pixel in several frames, walk the same nodes
        // (4096 spheres, 960 x 540, 4 frames: 8 x 8 pixels 31.3 ms, 4 x 8 22.5, 4 x 4 19.9, 2 x 4 19.4,
        // 2 x 2 20.2, 1 x 1 20.5).  Launches of many frames do better with more frames of the same few pixels
        // in a unit (8 pixels x 8 or 16 frames): 4096 spheres, 1920 x 1080, depth 50, units of 32 / 64 / 128 / 256
        // items: 8 frames 97.8 / 95.2 / 96.3 / 101.7 ms, 16 frames 191.1 / 184.5 / 183.8 / 187.5 ms, 32 frames
        // 378.0 / 363.8 / 359.9 / 361.3 ms.
        uint32_t target_items = args.tree_nodes != 0 ? (args.n_frames >= 16u ? 128u : (args.n_frames >= 8u ? 64u : 32u)) : 512u;
        if (args.tree_nodes == 0) {
            // A launch ends with a drain phase about as long as one unit (1 to 2 ms for 512 items of the
            // benchmark), which a launch of 30 rounds of units does not notice and a launch of 4 rounds
            // (one of eight shards of the image, say) pays with a third of its time.  Launches of fewer
            // than 16 rounds of 512-item units get smaller units, down to 128 items (two per lane).
            // Measured on the benchmark: the whole image 29.9 / 30.4 / 31.3 ms with units of 512 / 256 /
            // 128 items, the eight stripe shards of it together 41.8 / 36.2 / 34.5 ms.
            const uint64_t items = (uint64_t)args.width * rows * args.n_frames;
            const uint64_t slots = resident_wave_slots() * 16u;
            while (target_items > 128u && items / target_items < slots) {
                target_items >>= 1;
            }
        }
        if (args.unit_items != 0u) {
            target_items = args.unit_items;
        }
        // (up to 8 x 16 pixels where the host asks for it: single-frame launches that overlap their neighbours,
        // whose waves otherwise hold one item per lane and wait for the slowest)
        uint32_t log2_pixels = args.max_tile_log2 >= 3u && args.max_tile_log2 <= 8u ? args.max_tile_log2 : 6u;
        while (log2_pixels > 3u && (args.frames_per_unit << log2_pixels) > target_items) {
            --log2_pixels;
        }
        if ((args.frames_per_unit << log2_pixels) > target_items) {
            args.frames_per_unit = std::max(1u, target_items >> log2_pixels); // the smallest tile: fewer frames per unit
        }
        args.tile_h_log2 = (log2_pixels + 1u) / 2u;
        args.tile_w_log2 = log2_pixels - args.tile_h_log2;
    }
    // a tile must not straddle two stripes: its height divides stripe_rows (wider instead, up to 8)
    while (args.stripe_rows != 0u && args.stripe_rows % (1u << args.tile_h_log2) != 0u) {
        --args.tile_h_log2;
        if (args.tile_w_log2 < 4u) {
            ++args.tile_w_log2;
        }
    }
    const uint32_t tile_w = 1u << args.tile_w_log2, tile_h = 1u << args.tile_h_log2;
    args.tiles_x = (args.width + tile_w - 1u) / tile_w;
    const uint32_t tiles_y = (rows + tile_h - 1u) / tile_h;
    const uint64_t tiles = (uint64_t)args.tiles_x * tiles_y;
    args.n_chunks = (args.n_frames + args.frames_per_unit - 1) / args.frames_per_unit;
    const uint64_t units = tiles * args.n_chunks;
    if (units > 0x7FFFFFFFull) {
        return hipErrorInvalidValue;
    }
    args.n_tiles = (uint32_t)tiles;
    if (args.n_frames > 1 && args.partial == nullptr) {
        return hipErrorInvalidValue; // several frames need the per-frame sum image
    }
    if (args.tree_nodes != 0) {
        if (args.unit_counter == nullptr) {
            return hipErrorInvalidValue;
        }
        const hipError_t zeroed = hipMemsetAsync(args.unit_counter, 0, sizeof(uint32_t), stream);
        if (zeroed != hipSuccess) {
            return zeroed;
        }
    }
    if (args.tree_nodes != 0) {
        if (args.tree_packed == kTreeFormatQuadGlobal) {
            if (args.tree_geom == nullptr || args.quad_first_leaf >= args.tree_nodes || args.quad_roots == 0u ||
                args.quad_roots > kQuadMaxRoots || args.quad_prologue > 4u) {
                return hipErrorInvalidValue;
            }
            launch_trace_as<kTreeQuadGlobal, 256, kByCounter>(args, units, 4u * kWaveTableWords * 4u, stream);
            return hipGetLastError();
        }
        if (args.tree_packed == kTreeFormatQuad) {
            // 96 bytes per box group, 64 per group of spheres; the block size follows the footprint as below
            const size_t groups_bytes = (size_t)args.quad_first_leaf * 96u + ((size_t)args.tree_nodes - args.quad_first_leaf) * 64u;
            if (args.tree_geom == nullptr || args.quad_first_leaf >= args.tree_nodes || groups_bytes > kQuadMaxLdsBytes ||
                args.quad_roots == 0u || args.quad_roots > kQuadMaxRoots || args.quad_prologue > 4u) {
                return hipErrorInvalidValue;
            }
            if (groups_bytes <= 30u * 1024u) {
                launch_trace_as<kTreeQuad, 256, kByCounter>(args, units, groups_bytes + 4u * kWaveTableWords * 4u, stream);
            } else if (groups_bytes <= 72u * 1024u) {
                launch_trace_as<kTreeQuad, 512, kByCounter>(args, units, groups_bytes + 8u * kWaveTableWords * 4u, stream);
            } else {
                launch_trace_as<kTreeQuad, 1024, kByCounter>(args, units, groups_bytes + 16u * kWaveTableWords * 4u, stream);
            }
            return hipGetLastError();
        }
        if (args.tree_geom == nullptr || args.tree_word == nullptr || (args.tree_packed != 0 && args.tree_nodes > kTreeMaxNodes)) {
            return hipErrorInvalidValue;
        }
        if (args.tree_packed == 0) {
            launch_trace_as<kTreeGlobal, 256, kByCounter>(args, units, 4u * kWaveTableWords * 4u, stream);
            return hipGetLastError();
        }
        // a float4 and a word per node; the block size follows the footprint so that a CU's 160 KiB
        // of LDS still hold 16 to 20 waves
        const size_t nodes_bytes = (size_t)args.tree_nodes * 20u;
        if (nodes_bytes <= 30u * 1024u) {
            launch_trace_as<kTree, 256, kByCounter>(args, units, nodes_bytes + 4u * kWaveTableWords * 4u, stream);
        } else if (nodes_bytes <= 72u * 1024u) {
            launch_trace_as<kTree, 512, kByCounter>(args, units, nodes_bytes + 8u * kWaveTableWords * 4u, stream);
        } else {
            launch_trace_as<kTree, 1024, kByCounter>(args, units, nodes_bytes + 16u * kWaveTableWords * 4u, stream);
        }
        return hipGetLastError();
    }
    if (args.params.frame_sample_depth_count[3] > kMaxSpheresGeomLds) {
        if (args.loop_geom == nullptr) {
            return hipErrorInvalidValue;
        }
        launch_trace_as<kGeomGlobal, 256, kOneUnit>(args, units, 4u * kWaveTableWords * 4u, stream);
        return hipGetLastError();
    }
    bool full = false;
    const size_t lds = trace_lds_bytes(args.params.frame_sample_depth_count[3], &full);
    if (full) {
        // One wave per workgroup for scenes whose staging is a handful of records: a finished wave is
        // replaced at once instead of waiting for its three siblings (+1 % on the benchmark).
        if (args.params.frame_sample_depth_count[3] <= 16u) {
            if (args.params.frame_sample_depth_count[3] == 5u) {
                launch_trace_as<kFullLds5, 64, kOneUnit>(args, units, lds, stream);
            } else {
                launch_trace_as<kFullLds, 64, kOneUnit>(args, units, lds, stream);
            }
        } else {
            launch_trace_as<kFullLds, 256, kOneUnit>(args, units, lds, stream);
        }
    } else {
        launch_trace_as<kGeomLds, 512, kOneUnit>(args, units, lds, stream);
    }
    return hipGetLastError();
}

hipError_t launch_fold(float4* accum, const float4* partial, uint32_t* output, uint32_t width, uint32_t height,
                       const RowSet& rows, uint32_t n_frames, float sample_count, hipStream_t stream)
{
    const uint64_t n_rows = rows.stripe_rows != 0u ? (uint64_t)rows.n_stripes * rows.stripe_rows : rows.row_end - rows.row_begin;
    const uint64_t count = n_rows * width;
    if (count == 0 || n_frames == 0) {
        return hipSuccess;
    }
    if (count > 0xFFFFFFFFull) {
        return hipErrorInvalidValue;
    }
    const uint32_t blocks = (uint32_t)((count + kBlockThreads - 1) / kBlockThreads);
    hipLaunchKernelGGL(rt_fold_kernel, dim3(blocks), dim3(kBlockThreads), 0, stream, accum, partial, output,
                       (size_t)width * height, width, height, rows, (uint32_t)count, n_frames, sample_count);
    return hipGetLastError();
}

int stats_section_count() { return ST_COUNT; }
int stats_phase_count() { return PH_COUNT; }

hipError_t launch_resolve(const float4* accum, uint32_t* output, uint32_t first, uint32_t count, float sample_count, hipStream_t stream)
{
    if (count == 0) {
        return hipSuccess;
    }
    const uint32_t blocks = (count + kBlockThreads - 1) / kBlockThreads;
    hipLaunchKernelGGL(rt_resolve_kernel, dim3(blocks), dim3(kBlockThreads), 0, stream, accum, output, first, count, sample_count);
    return hipGetLastError();
}

hipError_t launch_probe_hit(const rt_gpu_sphere* spheres, const float* rays, uint32_t n, float* out, int32_t* front, hipStream_t stream)
{
    hipLaunchKernelGGL(rt_probe_hit_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, spheres, rays, n, out, front);
    return hipGetLastError();
}

hipError_t launch_probe_pow(const float* x, const float* y, uint32_t n, float* out, hipStream_t stream)
{
    hipLaunchKernelGGL(rt_probe_pow_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, x, y, n, out);
    return hipGetLastError();
}

hipError_t launch_probe_arith(uint32_t op, const float* a, const float* b, uint32_t n, float* out, hipStream_t stream)
{
    hipLaunchKernelGGL(rt_probe_arith_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, op, a, b, n, out);
    return hipGetLastError();
}

hipError_t launch_probe_rng(const uint32_t* xyz, uint32_t n, uint32_t draws, uint32_t* out, hipStream_t stream)
{
    hipLaunchKernelGGL(rt_probe_rng_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, xyz, n, draws, out);
    return hipGetLastError();
}

} // namespace rt

// Launch interface between the host layer (rt_host.cpp) and the HIP kernels
// (rt_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/rt_mi355x.h"

namespace rt {

// Largest sphere count whose full 48-byte records are staged in LDS.
constexpr uint32_t kMaxSpheresFullLds = 1024;
// Largest sphere count whose 16-byte geometry records are staged in LDS (64 KiB) when every ray
// tests every sphere; beyond it that loop reads them from global memory.
constexpr uint32_t kMaxSpheresGeomLds = 4096;
// Largest sphere count the tracer accepts.
constexpr uint32_t kMaxSpheres = 1u << 20;

constexpr uint32_t kTreeFormatGlobal = 0, kTreeFormatPacked = 1, kTreeFormatQuad = 2, kTreeFormatQuadGlobal = 3;

// Kernel arguments of one trace launch.  `params` is the reference's uniform
// block (raytrace.comp.glsl:19-31) passed by value, so it lands in scalar
// registers; params.frame_sample_depth_count[0] is ignored in favour of
// first_frame/frame_stride.
struct TraceArgs {
    rt_gpu_params params;
    float4* accum;                // binding 0, float4 per pixel, pitch = width
    uint32_t* output;             // binding 1, rgba8 per pixel, or nullptr to skip the display tail
    const rt_gpu_sphere* spheres; // binding 2
    uint32_t* rng_out;            // optional per-pixel final RNG state
    float4* partial;              // [n_frames][height*width] per-frame sums; required when n_frames > 1
    uint32_t* stats;              // diagnostic builds only
    const float4* tree_geom;      // hierarchy nodes (rt_tree.hpp), or nullptr
    const uint32_t* tree_word;    // one packed word per node, or two words per node (tree_packed == 0)
    const float4* loop_geom;      // (centre, radius^2) per sphere in global memory, for scenes beyond kMaxSpheresGeomLds without a hierarchy
    uint32_t* unit_counter;       // one word the hierarchy kernel hands out its units from (zeroed by launch_trace)
    uint32_t tree_nodes;          // 0 = test every sphere; nodes of the depth-first tree, or groups of the four-way tree
    uint32_t tree_packed;         // kTreeFormat*: 1 = the depth-first tree fits LDS and the packed word; 0 = it is walked in
                                  // global memory, and both node arrays hold one more (unused) entry than tree_nodes;
                                  // 2 = the complete four-way tree (rt_tree.hpp: QuadTree) in LDS: tree_geom holds six
                                  // float4 per box group, then four per group of spheres, tree_word is unused, and
                                  // `spheres` is the copy of the sphere list in slot order (one record per slot of the
                                  // last level and the prologue, the sphere's index in the bits of albedo[3],
                                  // 0xFFFFFFFF for an unused slot); 3 = the same tree walked in global memory
    uint32_t quad_roots;          // four-way tree: groups on level 0,
    uint32_t quad_first_leaf;     // index of the first group of the last level (= number of box groups),
    uint32_t quad_prologue;       // spheres in the group behind the tree (index tree_nodes - 1), 0..4,
    float quad_margin;            // 26 * 2^-24 / smallest radius in the tree,
    float4 quad_bound;            // and a ball (centre, radius) around the spheres of the tree
    uint32_t scene_bounded;       // every sphere finite with |centre| <= 2^20 and |radius| <= 2^20
    uint32_t width;
    uint32_t height;
    uint32_t row_begin;
    uint32_t row_end;
    // Interleaved rows (stripe_rows != 0): the image is cut into stripes of stripe_rows rows and the
    // launch renders stripes stripe_first, stripe_first + stripe_stride, ... (n_stripes of them);
    // row_begin/row_end are then ignored.  A stripe cut short by the bottom of the image is allowed.
    uint32_t stripe_rows;
    uint32_t stripe_first;
    uint32_t stripe_stride;
    uint32_t n_stripes;
    uint32_t first_frame;
    uint32_t n_frames;
    uint32_t frame_stride;
    uint32_t frames_per_unit;     // frames of one tile a wave pools together (1..64)
    uint32_t tile_w_log2;         // pixel tile of a unit, 2^w x 2^h with w, h <= 4; anything larger = let launch_trace choose
    uint32_t tile_h_log2;
    uint32_t unit_items;          // items a unit should pool (0 = launch_trace's default)
    uint32_t max_tile_log2;       // log2 of the largest pixel tile launch_trace may choose (0 = 6, i.e. 8 x 8)
    // filled in by launch_trace
    uint32_t tiles_x;
    uint32_t n_tiles;
    uint32_t n_chunks;
};

// The file-scope device array behind rt_render_frames_static: kStaticScratchHeader bytes for the unit
// counter, then room for per-frame sums.  Each device holds its own copy for the life of the process.
constexpr size_t kStaticScratchHeader = 256;
constexpr size_t kStaticScratchBytes = kStaticScratchHeader + (static_cast<size_t>(64) << 20);
// Device address and size of that array on `device`, which must be the current device; looked up once
// per device.
hipError_t static_scratch(int device, void** ptr, size_t* bytes);

// Waves the current device holds at a time (32 per compute unit).
uint64_t resident_wave_slots();
size_t trace_lds_bytes(uint32_t sphere_count, bool* mat_in_lds);
hipError_t launch_trace(const TraceArgs& args, hipStream_t stream);
// Rows a launch covers: a contiguous band, or every stripe_stride-th stripe of stripe_rows rows.
struct RowSet {
    uint32_t row_begin = 0;
    uint32_t row_end = 0;
    uint32_t stripe_rows = 0; // 0 = the band [row_begin, row_end)
    uint32_t stripe_first = 0;
    uint32_t stripe_stride = 1;
    uint32_t n_stripes = 0;
};
hipError_t launch_fold(float4* accum, const float4* partial, uint32_t* output, uint32_t width, uint32_t height,
                       const RowSet& rows, uint32_t n_frames, float sample_count, hipStream_t stream);
int stats_section_count();
int stats_phase_count();
hipError_t launch_resolve(const float4* accum, uint32_t* output, uint32_t first, uint32_t count,
                          float sample_count, hipStream_t stream);
hipError_t launch_probe_hit(const rt_gpu_sphere* spheres, const float* rays, uint32_t n, float* out,
                            int32_t* front, hipStream_t stream);
hipError_t launch_probe_pow(const float* x, const float* y, uint32_t n, float* out, hipStream_t stream);
hipError_t launch_probe_arith(uint32_t op, const float* a, const float* b, uint32_t n, float* out, hipStream_t stream);
hipError_t launch_probe_rng(const uint32_t* xyz, uint32_t n, uint32_t draws, uint32_t* out, hipStream_t stream);

} // namespace rt

// Host layer of the MI355X sphere path tracer: the C ABI of include/rt_mi355x.h.
//
// It keeps the state the reference keeps in class RayTracer
// (src/rt/RayTracer.h:79-93): extent, reset flag, sphere list, camera, aperture,
// field of view, focus distance, samples per frame and bounce budget; builds the
// uniform block the way RayTracer::makeCameraParams does
// (src/rt/RayTracer.cpp:140-177); and replaces the Vulkan recording of
// RayTracer::render (:559-685) by a memset plus one kernel launch on a HIP stream.
//
// Compile with -ffp-contract=off: the camera block must come out with the bits
// the reference's GLM code produces.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/rt_mi355x.h"
#include "rt_kernels.hpp"
#include "rt_tree.hpp"

static_assert(sizeof(rt_gpu_sphere) == 48, "GPUSphere is 48 bytes (std430 stride)");
static_assert(sizeof(rt_gpu_params) == 160, "GPUParams is 160 bytes (std140)");
static_assert(offsetof(rt_gpu_params, frame_sample_depth_count) == 112, "uvec4 sits at byte 112");
static_assert(offsetof(rt_gpu_params, resolution) == 128, "resolution sits at byte 128");
static_assert(offsetof(rt_gpu_params, inv_resolution) == 144, "invResolution sits at byte 144");

#ifndef RT_PIPE_SLOTS
#define RT_PIPE_SLOTS 2
#endif

struct rt_tracer {
    int device = 0;
    uint32_t width = 0;
    uint32_t height = 0;

    float4* accum = nullptr;          // mAccumImage
    uint32_t* output = nullptr;       // the swapchain image the shader stores into
    rt_gpu_sphere* d_spheres = nullptr; // mSphereBuffer
    uint32_t d_sphere_capacity = 0;
    uint32_t* rng_state = nullptr;    // test hook
    bool rng_capture = false;
    float4* partial = nullptr;        // per-frame sums of a multi-frame launch: [frames][height*width]
    size_t partial_bytes = 0;
    float4* partial2 = nullptr;       // the same for the second slot of overlapping small multi-frame launches
    size_t partial2_bytes = 0;
    uint32_t* stats = nullptr;        // diagnostic builds only
    rt::SphereTree tree;              // hierarchy over `spheres` (empty for small or out-of-range scenes)
    rt::QuadTree quad;                // the same as a complete four-way tree, when that fits LDS (else empty)
    bool use_tree = true;
    bool use_quad = true;             // walk the four-way tree of boxes (false: the depth-first tree of balls)
    bool tree_on_device = false;      // d_tree_geom / d_tree_word hold `tree` of the current scene
    float4* d_quad_geom = nullptr;    // the four-way tree as the kernel stages it: six float4 per box group, then four per group of spheres
    rt_gpu_sphere* d_slot_spheres = nullptr; // the sphere list in slot order of the four-way tree
    size_t d_quad_capacity = 0;       // bytes
    uint32_t d_slot_capacity = 0;     // slots
    bool scene_bounded = false;       // every sphere finite, |centre| and |radius| at most 2^20
    float4* d_tree_geom = nullptr;
    uint32_t* d_tree_word = nullptr;
    uint32_t* d_unit_counter = nullptr;
    float4* d_loop_geom = nullptr;
    uint32_t d_loop_geom_capacity = 0;
    uint32_t d_tree_capacity = 0;

    bool reset_accum = true;          // mResetAccum
    std::vector<rt_gpu_sphere> spheres; // mSpheres

    // RayTracer.h:87-93
    float cam_pos[3] = { 13.0f, 2.0f, 3.0f };
    float cam_dir[3] = { -1.0f, 0.0f, 0.0f };
    float aperture = 0.05f;
    float vertical_fov = 20.0f;
    float focus_distance = 10.0f;
    uint32_t samples_per_pixel = 4;
    uint32_t max_depth = 12;

    // Overlap of consecutive single-frame calls (how the reference's caller drives render,
    // src/core/App.cpp:403): call k traces its frame on the library's own stream k % 2 into a per-frame
    // sum image of its own, and the sums are folded into the accumulator on the caller's stream, in call
    // order.  The drain phase of one launch then runs next to the start of the following one.
    bool overlap_frames = true;
    static constexpr uint32_t kPipeSlots = RT_PIPE_SLOTS;
    hipStream_t pipe_stream[kPipeSlots] = {};
    hipEvent_t pipe_traced[kPipeSlots] = {};   // the trace of the slot has finished
    hipEvent_t pipe_free[kPipeSlots] = {};     // the fold that read the slot has finished
    bool pipe_used[kPipeSlots] = {};
    float4* pipe_partial = nullptr;   // kPipeSlots images of per-frame sums
    uint64_t pipe_calls = 0;
    uint32_t pipe_last_slot = 0;      // slot of the most recent overlapped launch

    // Recorded on the caller's stream at the end of every render call: what a scene change or a growing
    // work buffer waits for before it frees memory that the handle's kernels read, instead of idling
    // the whole device (other handles, copies and collectives on the same GPU keep running).
    hipEvent_t last_render = nullptr;
    bool rendered = false;

    // HIP event pairs around the most recent trace launches (a ring), for rt_kernel_ms
    static constexpr uint32_t kTimedLaunches = 64;
    hipEvent_t ev_start[kTimedLaunches] = {};
    hipEvent_t ev_stop[kTimedLaunches] = {};
    uint64_t launches = 0;

    std::string error;
};

namespace {

thread_local std::string g_create_error;

// Mirrors VK_CHECK (src/util/Check.h:8-19): "<Result>(n) for <expr> at file:line".
int fail_hip(rt_tracer* h, hipError_t e, const char* expr, const char* file, int line)
{
    char buf[512];
    std::snprintf(buf, sizeof(buf), "hipError(%d: %s) for %s at %s:%d", (int)e, hipGetErrorName(e), expr, file, line);
    if (h != nullptr) {
        h->error = buf;
    } else {
        g_create_error = buf;
    }
    (void)hipGetLastError(); // clear the sticky per-thread error
    return e == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP;
}

// One handle per device overlaps its launches at a time: launches of different handles (on different streams of
// the application) run beside each other anyway, and the library's extra streams then only add contention
// (measured with 2 / 3 / 4 tracers rendering single frames on streams of their own: 0.087 / 0.068 / 0.055 ms per
// frame and tracer with one overlapping handle, 0.092 / 0.081 / 0.087 ms with all of them overlapping).  The first
// handle to make an overlapped call on a device owns the overlap until it is destroyed, or has it switched off
// while none of its overlapped launches is in flight.
constexpr int kMaxOverlapDevices = 64;
std::atomic<rt_tracer*> g_overlap_owner[kMaxOverlapDevices];

bool claim_overlap(rt_tracer* h)
{
    if (h->device < 0 || h->device >= kMaxOverlapDevices) {
        return false;
    }
    rt_tracer* owner = nullptr;
    return g_overlap_owner[h->device].compare_exchange_strong(owner, h) || owner == h;
}

void release_overlap(rt_tracer* h)
{
    if (h->device >= 0 && h->device < kMaxOverlapDevices) {
        rt_tracer* owner = h;
        (void)g_overlap_owner[h->device].compare_exchange_strong(owner, nullptr);
    }
}

// Blocks until the kernels of the handle's render calls so far have finished.
hipError_t wait_for_renders(rt_tracer* h)
{
    if (!h->rendered || h->last_render == nullptr) {
        return hipSuccess;
    }
    // A call that failed between its launch on one of the library's streams and the wait it queues on the
    // caller's stream leaves a kernel that `last_render` does not cover.  The library's streams never wait
    // on the device, so waiting for them here cannot stall.
    for (hipStream_t side : h->pipe_stream) {
        if (side != nullptr && hipStreamSynchronize(side) != hipSuccess) {
            (void)hipGetLastError();
            return hipDeviceSynchronize();
        }
    }
    if (hipEventSynchronize(h->last_render) == hipSuccess) {
        return hipSuccess;
    }
    // e.g. the event was recorded into a stream capture: it cannot be waited for, so wait for everything
    (void)hipGetLastError();
    return hipDeviceSynchronize();
}

// Records rt_tracer::last_render when a render call returns, whichever way it returns.
struct RenderMark {
    rt_tracer* h;
    hipStream_t stream;
    ~RenderMark()
    {
        if (h->last_render != nullptr && hipEventRecord(h->last_render, stream) == hipSuccess) {
            h->rendered = true;
        } else {
            (void)hipGetLastError();
        }
    }
};

int fail_arg(rt_tracer* h, const char* what)
{
    if (h != nullptr) {
        h->error = std::string("bad argument: ") + what;
    } else {
        g_create_error = std::string("bad argument: ") + what;
    }
    return RT_ERR_BAD_ARGUMENT;
}

// Test hook (rt_debug_fail_allocation): the n-th device allocation from now on reports
// hipErrorOutOfMemory, so that the error paths of resize / scene changes can be exercised.
std::atomic<int> g_fail_allocation_in{ 0 };
hipError_t device_malloc(void** ptr, size_t bytes)
{
    if (g_fail_allocation_in.load() > 0 && g_fail_allocation_in.fetch_sub(1) == 1) {
        *ptr = nullptr;
        return hipErrorOutOfMemory;
    }
    return hipMalloc(ptr, bytes);
}

#define RT_HIP_TRY(h, expr)                                              \
    do {                                                                 \
        hipError_t rt_e_ = (expr);                                       \
        if (rt_e_ != hipSuccess) {                                       \
            return fail_hip((h), rt_e_, #expr, __FILE__, __LINE__);      \
        }                                                                \
    } while (0)

// Makes the handle's device current for the duration of one call and puts the
// caller's device back afterwards.
class DeviceScope {
public:
    explicit DeviceScope(int device)
    {
        if (hipGetDevice(&previous_) != hipSuccess) {
            previous_ = -1;
        }
        if (previous_ != device) {
            status_ = hipSetDevice(device);
            switched_ = (status_ == hipSuccess);
        }
    }
    ~DeviceScope()
    {
        if (switched_ && previous_ >= 0) {
            (void)hipSetDevice(previous_);
        }
    }
    hipError_t status() const { return status_; }

private:
    int previous_ = -1;
    bool switched_ = false;
    hipError_t status_ = hipSuccess;
};

// ---- GLM 1.0.1 arithmetic used by RayTracer.cpp, restated in plain floats ----
struct V3 {
    float x, y, z;
};
inline V3 v3(float x, float y, float z) { return V3{ x, y, z }; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline float dot3(V3 a, V3 b)
{
    // glm::dot: tmp = a * b; tmp.x + tmp.y + tmp.z
    float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
    return tx + ty + tz;
}
inline V3 normalize3(V3 a)
{
    // glm::normalize = v * inversesqrt(dot(v, v)); glm::inversesqrt = 1 / sqrt
    float inv = 1.0f / std::sqrt(dot3(a, a));
    return a * inv;
}
inline V3 cross3(V3 a, V3 b)
{
    return v3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}

void set_sphere(rt_gpu_sphere& s, float cx, float cy, float cz, float r, float ar, float ag, float ab,
                float mat, float fuzz, float ior, float flags)
{
    s.center_radius[0] = cx; s.center_radius[1] = cy; s.center_radius[2] = cz; s.center_radius[3] = r;
    s.albedo[0] = ar; s.albedo[1] = ag; s.albedo[2] = ab; s.albedo[3] = 0.0f;
    s.misc[0] = mat; s.misc[1] = fuzz; s.misc[2] = ior; s.misc[3] = flags;
}

// RayTracer::buildScene, src/rt/RayTracer.cpp:105-138
void build_default_scene(std::vector<rt_gpu_sphere>& out)
{
    out.assign(5, rt_gpu_sphere{});
    set_sphere(out[0], 0.0f, -1000.0f, 0.0f, 1000.0f, 0.75f, 0.8f, 0.9f, 0.0f, 0.0f, 1.0f, 1.0f); // ground, checker
    set_sphere(out[1], 0.0f, 1.0f, 0.0f, 1.0f, 0.9f, 0.25f, 0.25f, 0.0f, 0.0f, 1.0f, 0.0f);       // centre, lambert
    set_sphere(out[2], -4.0f, 1.0f, 0.0f, 1.0f, 1.0f, 1.0f, 1.0f, 2.0f, 0.0f, 1.5f, 0.0f);        // left, glass
    set_sphere(out[3], 4.0f, 1.0f, 0.0f, 1.0f, 0.95f, 0.65f, 0.15f, 1.0f, 0.03f, 1.0f, 0.0f);     // right, fuzzy metal
    set_sphere(out[4], 2.5f, 0.5f, 2.5f, 0.5f, 0.95f, 0.95f, 0.98f, 1.0f, 0.0f, 1.0f, 0.0f);      // mirror
}

// RayTracer::makeCameraParams, src/rt/RayTracer.cpp:140-177
void make_camera_params(const rt_tracer& t, rt_gpu_params& params)
{
    const V3 look_from = v3(t.cam_pos[0], t.cam_pos[1], t.cam_pos[2]);
    const V3 direction = normalize3(v3(t.cam_dir[0], t.cam_dir[1], t.cam_dir[2]));
    const V3 look_at = look_from + direction;
    const V3 vup = v3(0.0f, 1.0f, 0.0f);

    const float aspect = static_cast<float>(t.width) / static_cast<float>(t.height);
    const float theta = t.vertical_fov * 0.01745329251994329576923690768489f; // glm::radians
    const float half_height = tanf(theta * 0.5f);
    const float viewport_height = 2.0f * half_height;
    const float viewport_width = aspect * viewport_height;

    const V3 w = normalize3(look_from - look_at);
    const V3 u = normalize3(cross3(vup, w));
    const V3 v = cross3(w, u);

    const float focus = t.focus_distance;
    const V3 horizontal = u * (focus * viewport_width);
    const V3 vertical = v * (focus * viewport_height);
    const V3 lower_left = look_from - horizontal * 0.5f - vertical * 0.5f - w * focus;

    std::memset(&params, 0, sizeof(params));
    params.origin_lens[0] = look_from.x; params.origin_lens[1] = look_from.y; params.origin_lens[2] = look_from.z;
    params.origin_lens[3] = t.aperture * 0.5f;
    params.lower_left[0] = lower_left.x; params.lower_left[1] = lower_left.y; params.lower_left[2] = lower_left.z;
    params.horizontal[0] = horizontal.x; params.horizontal[1] = horizontal.y; params.horizontal[2] = horizontal.z;
    params.vertical[0] = vertical.x; params.vertical[1] = vertical.y; params.vertical[2] = vertical.z;
    params.u[0] = u.x; params.u[1] = u.y; params.u[2] = u.z;
    params.v[0] = v.x; params.v[1] = v.y; params.v[2] = v.z;
    params.w[0] = w.x; params.w[1] = w.y; params.w[2] = w.z;
    params.resolution[0] = static_cast<float>(t.width);
    params.resolution[1] = static_cast<float>(t.height);
    params.inv_resolution[0] = 1.0f / static_cast<float>(t.width);
    params.inv_resolution[1] = 1.0f / static_cast<float>(t.height);
}

// RayTracer::updateParams, src/rt/RayTracer.cpp:550-557
void update_params(const rt_tracer& t, uint32_t frame_index, rt_gpu_params& params)
{
    make_camera_params(t, params);
    params.frame_sample_depth_count[0] = frame_index;
    params.frame_sample_depth_count[1] = t.samples_per_pixel;
    params.frame_sample_depth_count[2] = t.max_depth;
    params.frame_sample_depth_count[3] = static_cast<uint32_t>(t.spheres.size());
}

// The device images of one extent.
struct Images {
    float4* accum = nullptr;
    uint32_t* output = nullptr;
    uint32_t* rng_state = nullptr;
    void release()
    {
        if (accum) { (void)hipFree(accum); accum = nullptr; }
        if (output) { (void)hipFree(output); output = nullptr; }
        if (rng_state) { (void)hipFree(rng_state); rng_state = nullptr; }
    }
};

void free_images(rt_tracer* h)
{
    if (h->accum) { (void)hipFree(h->accum); h->accum = nullptr; }
    if (h->output) { (void)hipFree(h->output); h->output = nullptr; }
    if (h->rng_state) { (void)hipFree(h->rng_state); h->rng_state = nullptr; }
    if (h->partial) { (void)hipFree(h->partial); h->partial = nullptr; h->partial_bytes = 0; }
    if (h->partial2) { (void)hipFree(h->partial2); h->partial2 = nullptr; h->partial2_bytes = 0; }
    if (h->pipe_partial) { (void)hipFree(h->pipe_partial); h->pipe_partial = nullptr; }
    for (bool& used : h->pipe_used) { used = false; }
}

// RayTracer::createAccumulationImage, src/rt/RayTracer.cpp:498-527 (+ the output image
// that the reference borrows from the swapchain).  Allocates into `out`; the handle is
// not touched, so a failure leaves it as it was.
int allocate_images(rt_tracer* h, uint32_t width, uint32_t height, Images& out)
{
    const size_t pixels = static_cast<size_t>(width) * height;
    auto body = [&]() -> int {
        RT_HIP_TRY(h, device_malloc(reinterpret_cast<void**>(&out.accum), pixels * sizeof(float4)));
        RT_HIP_TRY(h, device_malloc(reinterpret_cast<void**>(&out.output), pixels * sizeof(uint32_t)));
        RT_HIP_TRY(h, hipMemset(out.output, 0, pixels * sizeof(uint32_t)));
        if (h->rng_capture) {
            RT_HIP_TRY(h, device_malloc(reinterpret_cast<void**>(&out.rng_state), pixels * sizeof(uint32_t)));
            RT_HIP_TRY(h, hipMemset(out.rng_state, 0, pixels * sizeof(uint32_t)));
        }
        return RT_OK;
    };
    const int rc = body();
    if (rc != RT_OK) {
        out.release();
    }
    return rc;
}

// What a sphere list needs besides its records: the range flag and the hierarchy.
struct SceneIndex {
    bool bounded = true;
    rt::SphereTree tree;
    rt::QuadTree quad;
};

// want_depth_first: also build the older depth-first tree of balls (needed only when the handle walks that tree;
// otherwise it is built when somebody asks for it, ensure_depth_first_tree).
void index_scene(const std::vector<rt_gpu_sphere>& spheres, bool want_depth_first, SceneIndex& out)
{
    out.bounded = true;
    for (const rt_gpu_sphere& s : spheres) {
        for (int k = 0; k < 4; ++k) {
            if (!(std::fabs(s.center_radius[k]) <= 0x1p20f)) { // also catches NaN
                out.bounded = false;
            }
        }
    }
    (void)rt::build_quad_tree(spheres.data(), static_cast<uint32_t>(spheres.size()), out.quad); // empty when the scene does not qualify
    if (want_depth_first && !out.quad.empty()) {
        (void)rt::build_sphere_tree(spheres.data(), static_cast<uint32_t>(spheres.size()), out.tree);
    }
}

// The depth-first tree as the kernels read it.  The node arrays hold one entry more than the tree has nodes: the
// walk fetches the node behind the one it is testing, also behind the last.  Words: one per node (packed) or two (wide).
void pack_depth_first_tree(const rt::SphereTree& tree, std::vector<float>& tree_geom, std::vector<uint32_t>& tree_words)
{
    const uint32_t nodes = tree.node_count();
    tree_geom.clear();
    tree_words.clear();
    if (nodes == 0) {
        return;
    }
    tree_geom = tree.geom;
    tree_geom.resize((static_cast<size_t>(nodes) + 1u) * 4u, 0.0f);
    if (tree.packed()) {
        tree_words = tree.word;
        tree_words.push_back(0u);
        // the LDS walk reads an inner node's continuation from its fourth float: sign bit set, the
        // value rounded up to a multiple of 2^13 ulps, the node index in the 13 bits that frees
        for (uint32_t i = 0; i < nodes; ++i) {
            if (tree.sphere[i] == rt::kTreeNoSphere) {
                uint32_t bits;
                std::memcpy(&bits, &tree_geom[4u * i + 3u], sizeof bits);
                bits = ((bits + 0x1FFFu) & ~0x1FFFu) | tree.link[i] | 0x80000000u;
                std::memcpy(&tree_geom[4u * i + 3u], &bits, sizeof bits);
            }
        }
    } else {
        tree_words.resize((static_cast<size_t>(nodes) + 1u) * 2u, 0u);
        for (uint32_t i = 0; i < nodes; ++i) {
            tree_words[2u * i] = tree.link[i] * 16u;
            tree_words[2u * i + 1u] = tree.sphere[i] == rt::kTreeNoSphere ? 0u : rt::kTreeLeafFlag | tree.sphere[i];
        }
    }
}

// A device buffer that is either the handle's current one (large enough) or a fresh allocation
// that replaces it once the whole scene change has succeeded.
struct Staged {
    void* ptr = nullptr;
    bool fresh = false;
    hipError_t take(void* current, size_t current_capacity, size_t need, size_t bytes)
    {
        if (current != nullptr && need <= current_capacity) {
            ptr = current;
            return hipSuccess;
        }
        fresh = true;
        return device_malloc(&ptr, bytes);
    }
    void drop() { if (fresh && ptr) { (void)hipFree(ptr); } ptr = nullptr; fresh = false; }
};

// RayTracer::uploadScene, src/rt/RayTracer.cpp:529-548, for a new sphere list: every device buffer the
// scene needs is allocated (where the current one is too small) and filled BEFORE the handle changes,
// so a failure half way leaves the previous scene, its hierarchy and its device copies intact.
int install_scene(rt_tracer* h, std::vector<rt_gpu_sphere>&& spheres)
{
    SceneIndex index;
    index_scene(spheres, !h->use_quad, index);
    if (h->device < 0) {
        h->spheres = std::move(spheres);
        h->tree = std::move(index.tree);
        h->tree_on_device = false;
        h->quad = std::move(index.quad);
        h->scene_bounded = index.bounded;
        return RT_OK;
    }
    const uint32_t n = static_cast<uint32_t>(spheres.size());
    const uint32_t nodes = index.tree.node_count();
    std::vector<float> tree_geom;
    std::vector<uint32_t> tree_words;
    pack_depth_first_tree(index.tree, tree_geom, tree_words);
    // the four-way tree: its groups, and the sphere records in slot order with the sphere's index in the unused
    // fourth albedo component (all ones for an unused slot)
    const uint32_t quad_groups = index.quad.group_count();
    const uint32_t quad_slots = index.quad.slot_count();
    const size_t quad_bytes = index.quad.lds_bytes();
    std::vector<rt_gpu_sphere> slot_spheres;
    std::vector<float> quad_image; // box groups, then the groups of spheres
    if (quad_groups != 0) {
        quad_image = index.quad.box;
        quad_image.insert(quad_image.end(), index.quad.geom.begin(), index.quad.geom.end());
        slot_spheres.assign(quad_slots, rt_gpu_sphere{});
        for (uint32_t slot = 0; slot < quad_slots; ++slot) {
            const uint32_t sphere = index.quad.slot_sphere[slot];
            if (sphere != rt::kTreeNoSphere) {
                slot_spheres[slot] = spheres[sphere];
            }
            std::memcpy(&slot_spheres[slot].albedo[3], &sphere, sizeof sphere);
        }
    }
    std::vector<float> loop_geom;
    if (n > rt::kMaxSpheresGeomLds) {
        // (centre, radius^2) records for the loop over all spheres in global memory
        loop_geom.resize(static_cast<size_t>(n) * 4u);
        for (uint32_t i = 0; i < n; ++i) {
            const float* cr = spheres[i].center_radius;
            loop_geom[4u * i + 0u] = cr[0];
            loop_geom[4u * i + 1u] = cr[1];
            loop_geom[4u * i + 2u] = cr[2];
            loop_geom[4u * i + 3u] = cr[3] * cr[3];
        }
    }

    // the previous scene may still be read by one of the handle's kernels in flight
    RT_HIP_TRY(h, wait_for_renders(h));
    Staged d_spheres, d_tree_geom, d_tree_word, d_loop_geom, d_counter, d_quad_geom, d_slot_spheres;
    auto body = [&]() -> int {
        RT_HIP_TRY(h, d_spheres.take(h->d_spheres, h->d_sphere_capacity, n, n * sizeof(rt_gpu_sphere)));
        if (nodes != 0) {
            RT_HIP_TRY(h, d_tree_geom.take(h->d_tree_geom, h->d_tree_capacity, nodes, (static_cast<size_t>(nodes) + 1u) * sizeof(float4)));
            RT_HIP_TRY(h, d_tree_word.take(h->d_tree_word, h->d_tree_capacity, nodes, (static_cast<size_t>(nodes) + 1u) * 2u * sizeof(uint32_t)));
        }
        if (nodes != 0 || quad_groups != 0) {
            RT_HIP_TRY(h, d_counter.take(h->d_unit_counter, 1, 1, 64u * (1u + rt_tracer::kPipeSlots)));
        }
        if (!loop_geom.empty()) {
            RT_HIP_TRY(h, d_loop_geom.take(h->d_loop_geom, h->d_loop_geom_capacity, n, static_cast<size_t>(n) * sizeof(float4)));
        }
        if (quad_groups != 0) {
            RT_HIP_TRY(h, d_quad_geom.take(h->d_quad_geom, h->d_quad_capacity, quad_bytes, quad_bytes));
            RT_HIP_TRY(h, d_slot_spheres.take(h->d_slot_spheres, h->d_slot_capacity, quad_slots, static_cast<size_t>(quad_slots) * sizeof(rt_gpu_sphere)));
        }
        return RT_OK;
    };
    int rc = body();
    if (rc == RT_OK) {
        // From here on buffers the handle already owns may be overwritten; if one of these blocking copies
        // fails the device copy of the scene is unusable, and the handle says so (render refuses to launch).
        auto copies = [&]() -> int {
            RT_HIP_TRY(h, hipMemcpy(d_spheres.ptr, spheres.data(), n * sizeof(rt_gpu_sphere), hipMemcpyHostToDevice));
            if (nodes != 0) {
                RT_HIP_TRY(h, hipMemcpy(d_tree_geom.ptr, tree_geom.data(), tree_geom.size() * sizeof(float), hipMemcpyHostToDevice));
                RT_HIP_TRY(h, hipMemcpy(d_tree_word.ptr, tree_words.data(), tree_words.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            }
            if (!loop_geom.empty()) {
                RT_HIP_TRY(h, hipMemcpy(d_loop_geom.ptr, loop_geom.data(), loop_geom.size() * sizeof(float), hipMemcpyHostToDevice));
            }
            if (quad_groups != 0) {
                RT_HIP_TRY(h, hipMemcpy(d_quad_geom.ptr, quad_image.data(), quad_image.size() * sizeof(float), hipMemcpyHostToDevice));
                RT_HIP_TRY(h, hipMemcpy(d_slot_spheres.ptr, slot_spheres.data(), slot_spheres.size() * sizeof(rt_gpu_sphere), hipMemcpyHostToDevice));
            }
            return RT_OK;
        };
        rc = copies();
        if (rc != RT_OK && !d_spheres.fresh && h->d_spheres != nullptr) {
            (void)hipFree(h->d_spheres);
            h->d_spheres = nullptr;
            h->d_sphere_capacity = 0;
            d_spheres.ptr = nullptr;
        }
    }
    if (rc != RT_OK) {
        d_spheres.drop();
        d_tree_geom.drop();
        d_tree_word.drop();
        d_loop_geom.drop();
        d_counter.drop();
        d_quad_geom.drop();
        d_slot_spheres.drop();
        return rc;
    }
    // commit
    if (d_spheres.fresh) {
        if (h->d_spheres) { (void)hipFree(h->d_spheres); }
        h->d_spheres = static_cast<rt_gpu_sphere*>(d_spheres.ptr);
        h->d_sphere_capacity = n;
    }
    if (d_tree_geom.fresh || d_tree_word.fresh) {
        // the two arrays share one capacity figure: both were (re)allocated together above
        if (h->d_tree_geom && d_tree_geom.fresh) { (void)hipFree(h->d_tree_geom); }
        if (h->d_tree_word && d_tree_word.fresh) { (void)hipFree(h->d_tree_word); }
        h->d_tree_geom = static_cast<float4*>(d_tree_geom.ptr);
        h->d_tree_word = static_cast<uint32_t*>(d_tree_word.ptr);
        h->d_tree_capacity = nodes;
    }
    if (d_counter.fresh) {
        h->d_unit_counter = static_cast<uint32_t*>(d_counter.ptr);
    }
    if (d_loop_geom.fresh) {
        if (h->d_loop_geom) { (void)hipFree(h->d_loop_geom); }
        h->d_loop_geom = static_cast<float4*>(d_loop_geom.ptr);
        h->d_loop_geom_capacity = n;
    }
    if (d_quad_geom.fresh) {
        if (h->d_quad_geom) { (void)hipFree(h->d_quad_geom); }
        h->d_quad_geom = static_cast<float4*>(d_quad_geom.ptr);
        h->d_quad_capacity = quad_bytes;
    }
    if (d_slot_spheres.fresh) {
        if (h->d_slot_spheres) { (void)hipFree(h->d_slot_spheres); }
        h->d_slot_spheres = static_cast<rt_gpu_sphere*>(d_slot_spheres.ptr);
        h->d_slot_capacity = quad_slots;
    }
    h->spheres = std::move(spheres);
    h->tree = std::move(index.tree);
    h->tree_on_device = nodes != 0;
    h->quad = std::move(index.quad);
    h->scene_bounded = index.bounded;
    return RT_OK;
}

// The depth-first tree of balls of the current scene, built when it is first asked for (the tracer walks the
// four-way tree of boxes; the older tree serves comparisons, rt_set_scene_hierarchy(h, 2), and its getters).
// on_device: also put it on the handle's device.  A failure leaves the handle as it was.
int ensure_depth_first_tree(rt_tracer* h, bool on_device)
{
    if (h->quad.empty()) {
        return RT_OK; // the scene has no hierarchy of either kind
    }
    rt::SphereTree built;
    const rt::SphereTree* tree = &h->tree;
    if (h->tree.empty()) {
        (void)rt::build_sphere_tree(h->spheres.data(), static_cast<uint32_t>(h->spheres.size()), built);
        tree = &built;
    }
    if (on_device && h->device >= 0 && !h->tree_on_device && !tree->empty()) {
        const uint32_t nodes = tree->node_count();
        std::vector<float> tree_geom;
        std::vector<uint32_t> tree_words;
        pack_depth_first_tree(*tree, tree_geom, tree_words);
        DeviceScope scope(h->device);
        RT_HIP_TRY(h, scope.status());
        RT_HIP_TRY(h, wait_for_renders(h)); // buffers of an earlier scene may be replaced
        Staged d_tree_geom, d_tree_word;
        auto body = [&]() -> int {
            RT_HIP_TRY(h, d_tree_geom.take(h->d_tree_geom, h->d_tree_capacity, nodes, (static_cast<size_t>(nodes) + 1u) * sizeof(float4)));
            RT_HIP_TRY(h, d_tree_word.take(h->d_tree_word, h->d_tree_capacity, nodes, (static_cast<size_t>(nodes) + 1u) * 2u * sizeof(uint32_t)));
            RT_HIP_TRY(h, hipMemcpy(d_tree_geom.ptr, tree_geom.data(), tree_geom.size() * sizeof(float), hipMemcpyHostToDevice));
            RT_HIP_TRY(h, hipMemcpy(d_tree_word.ptr, tree_words.data(), tree_words.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            return RT_OK;
        };
        const int rc = body();
        if (rc != RT_OK) {
            d_tree_geom.drop();
            d_tree_word.drop();
            return rc;
        }
        if (d_tree_geom.fresh || d_tree_word.fresh) {
            if (h->d_tree_geom && d_tree_geom.fresh) { (void)hipFree(h->d_tree_geom); }
            if (h->d_tree_word && d_tree_word.fresh) { (void)hipFree(h->d_tree_word); }
            h->d_tree_geom = static_cast<float4*>(d_tree_geom.ptr);
            h->d_tree_word = static_cast<uint32_t*>(d_tree_word.ptr);
            h->d_tree_capacity = nodes;
        }
        h->tree_on_device = true;
    }
    if (tree == &built) {
        h->tree = std::move(built);
    }
    return RT_OK;
}

// RayTracer::create, src/rt/RayTracer.cpp:188-193: default scene and the create-time aim
void init_default_state(rt_tracer* h)
{
    const V3 look_at = v3(0.0f, 1.0f, 0.0f);
    const V3 pos = v3(h->cam_pos[0], h->cam_pos[1], h->cam_pos[2]);
    const V3 d = look_at - pos;
    const V3 nd = normalize3(d);
    h->cam_dir[0] = nd.x; h->cam_dir[1] = nd.y; h->cam_dir[2] = nd.z;
    h->focus_distance = std::sqrt(dot3(d, d)); // glm::length
}

int fail_no_device(rt_tracer* h)
{
    h->error = "handle was created with RT_DEVICE_NONE: it holds host state only";
    return RT_ERR_NO_DEVICE;
}

#define RT_NEED_DEVICE(h)                  \
    do {                                   \
        if ((h)->device < 0) {             \
            return fail_no_device(h);      \
        }                                  \
    } while (0)

bool valid_extent(uint32_t w, uint32_t hgt)
{
    // the kernels index texels with 32 bits
    return w >= 1 && hgt >= 1 && w <= 65536u && hgt <= 65536u && static_cast<uint64_t>(w) * hgt < (1ull << 32);
}

// Clears the rows of `rows` in the accumulator (vkCmdClearColorImage of :595-600, restricted to a shard).
int clear_rows(rt_tracer* h, const rt::RowSet& rows, hipStream_t stream)
{
    const size_t row_bytes = static_cast<size_t>(h->width) * sizeof(float4);
    if (rows.stripe_rows == 0) {
        if (rows.row_end > rows.row_begin) {
            RT_HIP_TRY(h, hipMemsetAsync(h->accum + static_cast<size_t>(rows.row_begin) * h->width, 0,
                                         static_cast<size_t>(rows.row_end - rows.row_begin) * row_bytes, stream));
        }
        return RT_OK;
    }
    for (uint32_t k = 0; k < rows.n_stripes; ++k) {
        const uint64_t top = (static_cast<uint64_t>(rows.stripe_first) + static_cast<uint64_t>(k) * rows.stripe_stride) * rows.stripe_rows;
        if (top >= h->height) {
            break;
        }
        const uint64_t bottom = std::min<uint64_t>(top + rows.stripe_rows, h->height);
        RT_HIP_TRY(h, hipMemsetAsync(h->accum + top * h->width, 0, (bottom - top) * row_bytes, stream));
    }
    return RT_OK;
}

// static_scratch: per-frame sums and the unit counter live in the library's file-scope device array
// instead of allocations owned by the handle, and nothing is allocated or freed.
int render_impl(rt_tracer* h, uint32_t first_frame, uint32_t n_frames, uint32_t frame_stride,
                const rt::RowSet& rows, bool write_output, hipStream_t stream, bool static_scratch = false)
{
    if (rows.stripe_rows == 0) {
        if (rows.row_begin > rows.row_end || rows.row_end > h->height) {
            return fail_arg(h, "row range outside the image");
        }
    } else {
        const uint64_t stripes_in_image = (static_cast<uint64_t>(h->height) + rows.stripe_rows - 1) / rows.stripe_rows;
        if (rows.stripe_stride == 0) {
            return fail_arg(h, "stripe_stride must be >= 1");
        }
        if (rows.n_stripes != 0 &&
            static_cast<uint64_t>(rows.stripe_first) + static_cast<uint64_t>(rows.n_stripes - 1) * rows.stripe_stride >= stripes_in_image) {
            return fail_arg(h, "stripe range outside the image");
        }
    }
    RT_NEED_DEVICE(h);
    if (n_frames == 0) {
        return RT_OK;
    }
    if (frame_stride == 0) {
        return fail_arg(h, "frame_stride must be >= 1");
    }
    if (h->accum == nullptr || h->output == nullptr || h->d_spheres == nullptr) {
        // an earlier rt_resize / rt_set_spheres ran out of memory half way
        h->error = "the handle has no images or no scene on the device (an earlier resize or scene upload failed)";
        return RT_ERR_HIP;
    }
    DeviceScope scope(h->device);
    RT_HIP_TRY(h, scope.status());
    const RenderMark mark{ h, stream }; // destroyed before `scope`, so with the handle's device still current

    const size_t image_pixels = static_cast<size_t>(h->width) * h->height;
    const size_t frame_bytes = image_pixels * sizeof(float4);
    // a scene has both hierarchies or none; the depth-first one is walked only where it has been put on the device
    const bool depth_first = h->use_tree && !h->use_quad && !h->tree.empty() && h->tree_on_device;
    const bool quad = h->use_tree && !depth_first && !h->quad.empty() && h->d_quad_geom != nullptr && h->d_slot_spheres != nullptr;
    const bool tree = depth_first || quad;

    // Workspace of this call: per-frame sums (launches of several frames) and the unit counter (hierarchy).
    float4* partial = h->partial;
    uint32_t* unit_counter = h->d_unit_counter;
    uint32_t group = n_frames;
    if (static_scratch) {
        void* base = nullptr;
        size_t bytes = 0;
        RT_HIP_TRY(h, rt::static_scratch(h->device, &base, &bytes));
        unit_counter = static_cast<uint32_t*>(base);
        partial = reinterpret_cast<float4*>(static_cast<unsigned char*>(base) + rt::kStaticScratchHeader);
        const size_t room = bytes - rt::kStaticScratchHeader;
        if (n_frames > 1) {
            if (2 * frame_bytes > room) {
                char buf[200];
                std::snprintf(buf, sizeof(buf), "two frames of %ux%u need %zu bytes of per-frame sums, the static scratch holds %zu "
                              "(at most %zu pixels); use rt_render_frames", h->width, h->height, 2 * frame_bytes, room,
                              room / (2 * sizeof(float4)));
                return fail_arg(h, buf);
            }
            group = static_cast<uint32_t>(std::min<size_t>(n_frames, room / frame_bytes));
        }
    }

    // Launches may run beside their predecessor (see rt_tracer::overlap_frames): single frames always,
    // launches of several frames when they are small enough for the drain phase to matter (fewer than
    // 16 rounds of units: the shards of an image split over several GPUs, say).  Large launches stay on
    // the caller's stream.  Not while the final RNG words are captured: two launches in flight would both
    // store theirs.
    const uint64_t launch_rows = rows.stripe_rows != 0 ? static_cast<uint64_t>(rows.n_stripes) * rows.stripe_rows
                                                       : rows.row_end - rows.row_begin;
    const bool small_launch = launch_rows * h->width * n_frames / 512u < rt::resident_wave_slots() * 16u;
    // Not while the caller's stream is being captured into a HIP graph either: a launch on the library's
    // streams would run at once instead of becoming part of the graph, and the captured stream would be
    // told to wait for an event that is not part of the capture.  The call then stays on the caller's
    // stream as a whole and is captured like any other work.  (A query that fails -- the legacy stream while
    // another stream captures in global mode -- leaves the decision to the launches below, which fail the
    // same way.)
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &capture) != hipSuccess) {
        (void)hipGetLastError();
        capture = hipStreamCaptureStatusNone;
    }
    const bool capturing = capture != hipStreamCaptureStatusNone;
    bool overlap = !static_scratch && !capturing && h->overlap_frames && !h->rng_capture && h->stats == nullptr &&
                   (n_frames == 1 || small_launch) && claim_overlap(h);
    if (overlap && n_frames == 1 && h->pipe_partial == nullptr) {
        // The two images single-frame launches sum into (32 bytes per pixel) are allocated when the handle
        // overlaps for the first time: handles that never do (RNG capture, overlap switched off, the device's
        // overlap taken by another handle) do not pay for them.  If they do not fit, the call is not overlapped.
        if (device_malloc(reinterpret_cast<void**>(&h->pipe_partial), rt_tracer::kPipeSlots * frame_bytes) != hipSuccess) {
            (void)hipGetLastError();
            h->pipe_partial = nullptr;
            overlap = false;
        }
    }

    // A launch of several frames keeps one float4 per (frame, pixel) in HBM and folds them into the
    // accumulator in frame order afterwards; the frames of one call are split into groups so that
    // this image stays within a budget: 16 GiB (a 64-frame batch at 3840x2160 is 8.5 GB and should
    // stay one launch on a 288 GB device), at most a quarter of the memory that is free when the
    // buffer has to grow, or what RT_PARTIAL_BYTES says.  Overlapping launches alternate between two
    // such images.
    if (n_frames > 1 && !static_scratch) {
        auto ensure = [&](float4*& buffer, size_t& bytes) -> int {
            size_t budget = static_cast<size_t>(16) << 30;
            if (const char* e = std::getenv("RT_PARTIAL_BYTES")) {
                budget = static_cast<size_t>(std::strtoull(e, nullptr, 10));
            }
            if (static_cast<size_t>(group) * frame_bytes > bytes) {
                size_t free_bytes = 0, total_bytes = 0;
                if (hipMemGetInfo(&free_bytes, &total_bytes) == hipSuccess) {
                    budget = std::min(budget, std::max(bytes, (free_bytes + bytes) / 4));
                } else {
                    (void)hipGetLastError();
                }
            }
            const size_t fit = std::max<size_t>(2, budget / frame_bytes);
            group = static_cast<uint32_t>(std::min<size_t>(group, fit));
            if (static_cast<size_t>(group) * frame_bytes > bytes) {
                // kernels of earlier calls may still be reading the old buffer
                RT_HIP_TRY(h, wait_for_renders(h));
                if (buffer) {
                    (void)hipFree(buffer);
                    buffer = nullptr;
                    bytes = 0;
                }
                // an allocation that does not fit is retried with half the frames, down to two
                for (;;) {
                    const size_t need = static_cast<size_t>(group) * frame_bytes;
                    const hipError_t e = device_malloc(reinterpret_cast<void**>(&buffer), need);
                    if (e == hipSuccess) {
                        bytes = need;
                        break;
                    }
                    buffer = nullptr;
                    if (e != hipErrorOutOfMemory || group <= 2) {
                        return fail_hip(h, e, "hipMalloc(per-frame sums)", __FILE__, __LINE__);
                    }
                    (void)hipGetLastError();
                    group = std::max(2u, group / 2u);
                }
            }
            return RT_OK;
        };
        int rc = ensure(h->partial, h->partial_bytes);
        if (rc == RT_OK && overlap) {
            rc = ensure(h->partial2, h->partial2_bytes); // may shrink `group` further; the first buffer is then larger than needed
        }
        if (rc != RT_OK) {
            return rc;
        }
        partial = h->partial;
    }
    // RayTracer::render, src/rt/RayTracer.cpp:564-603.  A raised flag clears the whole image once;
    // frame 0 clears the rows this call renders (the whole image for an unsharded call), so row
    // bands rendered one after another on one handle do not wipe each other.
    if (h->reset_accum) {
        RT_HIP_TRY(h, hipMemsetAsync(h->accum, 0, frame_bytes, stream));
        h->reset_accum = false;
    } else if (first_frame == 0) {
        const int rc = clear_rows(h, rows, stream);
        if (rc != RT_OK) {
            return rc;
        }
    }

    // Unit geometry (frames pooled per unit, pixel tile): defaults are chosen by launch_trace;
    // RT_FRAMES_PER_UNIT and RT_TILE_LOG2="w h" override them for experiments.
    uint32_t frames_per_unit = 64;
    if (const char* e = std::getenv("RT_FRAMES_PER_UNIT")) {
        frames_per_unit = static_cast<uint32_t>(std::max(1, std::atoi(e)));
    }
    uint32_t tile_w_log2 = 0xFFu, tile_h_log2 = 0xFFu;
    if (const char* e = std::getenv("RT_TILE_LOG2")) {
        unsigned w = 0, hgt = 0;
        if (std::sscanf(e, "%u %u", &w, &hgt) == 2 && w <= 4 && hgt <= 4) {
            tile_w_log2 = w;
            tile_h_log2 = hgt;
        }
    }

    for (uint32_t done = 0; done < n_frames; done += group) {
        const uint32_t count = std::min(group, n_frames - done);
        const uint32_t first = first_frame + done * frame_stride;
        const bool last_group = done + count == n_frames;
        rt::TraceArgs args;
        update_params(*h, first, args.params);
        args.accum = h->accum;
        args.output = (write_output && count == 1) ? h->output : nullptr;
        args.spheres = h->d_spheres;
        args.rng_out = (h->rng_capture && last_group) ? h->rng_state : nullptr;
        args.partial = count > 1 ? partial : nullptr;
        args.stats = h->stats;
        args.tree_geom = depth_first ? h->d_tree_geom : nullptr;
        args.tree_word = depth_first ? h->d_tree_word : nullptr;
        args.unit_counter = tree ? unit_counter : nullptr;
        args.tree_nodes = depth_first ? h->tree.node_count() : 0u;
        args.tree_packed = depth_first && h->tree.packed() ? rt::kTreeFormatPacked : rt::kTreeFormatGlobal;
        args.quad_roots = args.quad_first_leaf = args.quad_prologue = 0u;
        args.quad_margin = 0.0f;
        args.quad_bound = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (quad) {
            args.tree_geom = h->d_quad_geom;
            args.tree_word = nullptr;
            args.tree_nodes = h->quad.group_count();
            static const bool quad_in_global = std::getenv("RT_QUAD_GLOBAL") != nullptr; // experiments: never stage the tree in LDS
            args.tree_packed = h->quad.lds_bytes() <= rt::kQuadMaxLdsBytes && !quad_in_global ? rt::kTreeFormatQuad : rt::kTreeFormatQuadGlobal;
            args.quad_roots = h->quad.roots;
            args.quad_first_leaf = h->quad.first_leaf;
            args.quad_prologue = h->quad.prologue_spheres;
            args.quad_bound = make_float4(h->quad.bound_centre[0], h->quad.bound_centre[1], h->quad.bound_centre[2], h->quad.bound_radius);
            args.quad_margin = 26.0f * 0x1p-24f / h->quad.min_radius;
            args.spheres = h->d_slot_spheres; // `best` of that walk is a slot
        }
        args.loop_geom = h->d_loop_geom;
        args.scene_bounded = h->scene_bounded ? 1u : 0u;
        args.width = h->width;
        args.height = h->height;
        args.row_begin = rows.row_begin;
        args.row_end = rows.row_end;
        args.stripe_rows = rows.stripe_rows;
        args.stripe_first = rows.stripe_first;
        args.stripe_stride = rows.stripe_stride;
        args.n_stripes = rows.n_stripes;
        args.first_frame = first;
        args.n_frames = count;
        args.frame_stride = frame_stride;
        args.frames_per_unit = frames_per_unit;
        args.tile_w_log2 = tile_w_log2;
        args.tile_h_log2 = tile_h_log2;
        args.tiles_x = args.n_tiles = args.n_chunks = 0;
        args.unit_items = 0;
        args.max_tile_log2 = 0;
        if (overlap && n_frames == 1 && !tree) {
            // Two items per lane (8 x 16 pixel tiles): a lane that finishes its pixel early takes another one
            // instead of waiting for the slowest lane of its wave.  Alone such a launch is slower (half as
            // many waves, a longer drain: 1.0 ms against 0.76 ms for the benchmark's frame); beside a
            // neighbour it is faster (0.53 against 0.55 ms; 0.195 against 0.205 ms at 4 spp).  So it is chosen
            // when the previous call's launch, on the library's other stream, has not finished yet.
            const uint32_t other = h->pipe_last_slot;
            bool neighbour_running = false;
            if (h->pipe_used[other]) {
                const hipError_t q = hipEventQuery(h->pipe_traced[other]);
                neighbour_running = q == hipErrorNotReady;
                if (q != hipSuccess) {
                    (void)hipGetLastError();
                }
            }
            if (neighbour_running) {
                args.unit_items = 128;
                args.max_tile_log2 = 7;
            }
        }
        if (const char* e = std::getenv("RT_UNIT_ITEMS")) {
            args.unit_items = static_cast<uint32_t>(std::max(1, std::atoi(e)));
        }

        const uint32_t slot = static_cast<uint32_t>(h->launches % rt_tracer::kTimedLaunches);
        if (overlap) {
            // this launch's samples are summed per pixel and frame into the call's slot on one of the library's
            // two streams (nothing there touches the accumulator, so it can run beside the previous call's
            // launch), and added to the accumulator on the caller's stream in frame order: prev + sum, as
            // :333-335 do
            // (launches of several frames alternate between the two per-frame-sum buffers of the handle)
            const uint32_t p = static_cast<uint32_t>(h->pipe_calls++ % (count == 1 ? rt_tracer::kPipeSlots : 2u));
            h->pipe_last_slot = p;
            hipStream_t side = h->pipe_stream[p];
            if (h->pipe_used[p]) {
                // The slot's previous sums must have been folded.  The HOST waits for that (it is the call before
                // last, so the launch of the last call keeps the GPU busy meanwhile), not the library's stream:
                // the library's streams then never wait for anything on the device, and no ring of streams
                // waiting for each other can form when the runtime lets several streams share a hardware queue
                // (which it does from the fifth active stream of a process on, and which stalled this
                // pipeline for good when it still waited on the device).
                RT_HIP_TRY(h, hipEventSynchronize(h->pipe_free[p]));
            }
            args.partial = count == 1 ? h->pipe_partial + static_cast<size_t>(p) * image_pixels : (p == 0 ? h->partial : h->partial2);
            args.output = nullptr;
            args.unit_counter = tree ? h->d_unit_counter + 16u * (1u + p) : nullptr;
            RT_HIP_TRY(h, hipEventRecord(h->ev_start[slot], side));
            RT_HIP_TRY(h, rt::launch_trace(args, side));
            RT_HIP_TRY(h, hipEventRecord(h->ev_stop[slot], side));
            ++h->launches;
            RT_HIP_TRY(h, hipEventRecord(h->pipe_traced[p], side));
            RT_HIP_TRY(h, hipStreamWaitEvent(stream, h->pipe_traced[p], 0));
            // (frame + 1u) * spp wraps in uint32 before the conversion, raytrace.comp.glsl:337
            const float sample_count = static_cast<float>((first + (count - 1) * frame_stride + 1u) * h->samples_per_pixel);
            RT_HIP_TRY(h, rt::launch_fold(h->accum, args.partial, (write_output && last_group) ? h->output : nullptr, h->width,
                                          h->height, rows, count, sample_count, stream));
            RT_HIP_TRY(h, hipEventRecord(h->pipe_free[p], stream));
            h->pipe_used[p] = true;
            continue;
        }
        RT_HIP_TRY(h, hipEventRecord(h->ev_start[slot], stream));
        RT_HIP_TRY(h, rt::launch_trace(args, stream));
        RT_HIP_TRY(h, hipEventRecord(h->ev_stop[slot], stream));
        ++h->launches;

        if (count > 1) {
            const uint32_t last_frame = first + (count - 1) * frame_stride;
            // (frame + 1u) * spp wraps in uint32 before the conversion, raytrace.comp.glsl:337
            const float sample_count = static_cast<float>((last_frame + 1u) * h->samples_per_pixel);
            RT_HIP_TRY(h, rt::launch_fold(h->accum, partial, (write_output && last_group) ? h->output : nullptr,
                                          h->width, h->height, rows, count, sample_count, stream));
            if (!static_scratch) {
                // `partial` is also slot 0 of the overlapping launches: one that follows on the library's own
                // stream must not write it before this fold has read it
                RT_HIP_TRY(h, hipEventRecord(h->pipe_free[0], stream));
                h->pipe_used[0] = true;
            }
        }
    }
    return RT_OK;
}

rt::RowSet band(uint32_t row_begin, uint32_t row_end)
{
    rt::RowSet rows;
    rows.row_begin = row_begin;
    rows.row_end = row_end;
    return rows;
}

} // namespace

extern "C" {

int rt_create(int device_id, uint32_t width, uint32_t height, rt_handle* out)
{
    if (out == nullptr) {
        return fail_arg(nullptr, "out is NULL");
    }
    *out = nullptr;
    if (!valid_extent(width, height)) {
        return fail_arg(nullptr, "extent must be within 1..65536");
    }
    if (device_id != RT_DEVICE_NONE) {
        int count = 0;
        hipError_t e = hipGetDeviceCount(&count);
        if (e != hipSuccess || count <= 0) {
            g_create_error = "no HIP device available";
            (void)hipGetLastError();
            return RT_ERR_NO_DEVICE;
        }
        if (device_id < 0 || device_id >= count) {
            return fail_arg(nullptr, "device_id out of range");
        }
    }
    rt_tracer* h = new (std::nothrow) rt_tracer();
    if (h == nullptr) {
        g_create_error = "host allocation failed";
        return RT_ERR_OUT_OF_MEMORY;
    }
    h->device = device_id;
    h->width = width;
    h->height = height;
    h->reset_accum = true;
    h->use_quad = std::getenv("RT_TREE_DEPTH_FIRST") == nullptr; // experiments: the one-node-per-step walk everywhere
    init_default_state(h);
    std::vector<rt_gpu_sphere> scene;
    build_default_scene(scene);
    if (device_id == RT_DEVICE_NONE) {
        // host-state-only handle: setters, rt_get_params and rt_get_spheres work, nothing renders
        (void)install_scene(h, std::move(scene));
        *out = h;
        return RT_OK;
    }

    int rc = RT_OK;
    {
        DeviceScope scope(device_id);
        auto body = [&]() -> int {
            RT_HIP_TRY(h, scope.status());
            for (uint32_t i = 0; i < rt_tracer::kTimedLaunches; ++i) {
                RT_HIP_TRY(h, hipEventCreate(&h->ev_start[i]));
                RT_HIP_TRY(h, hipEventCreate(&h->ev_stop[i]));
            }
            for (uint32_t i = 0; i < rt_tracer::kPipeSlots; ++i) {
                // Streams share a few hardware queues in turn (four per process by default), so with other
                // streams alive one of these can land on the queue of the caller's stream and run behind it
                // instead of beside it.  Measured: streams of the highest priority ran beside the caller's where
                // streams of normal priority queued behind it.
                int least = 0, greatest = 0;
                if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) {
                    (void)hipGetLastError();
                    greatest = 0;
                }
                const int priority = std::getenv("RT_PIPE_NORMAL_PRIORITY") != nullptr ? 0 : greatest;
                if (hipStreamCreateWithPriority(&h->pipe_stream[i], hipStreamNonBlocking, priority) != hipSuccess) {
                    (void)hipGetLastError(); // no priorities here: an ordinary stream still overlaps when a queue is free
                    RT_HIP_TRY(h, hipStreamCreateWithFlags(&h->pipe_stream[i], hipStreamNonBlocking));
                }
                RT_HIP_TRY(h, hipEventCreateWithFlags(&h->pipe_traced[i], hipEventDisableTiming));
                RT_HIP_TRY(h, hipEventCreateWithFlags(&h->pipe_free[i], hipEventDisableTiming));
            }
            RT_HIP_TRY(h, hipEventCreateWithFlags(&h->last_render, hipEventDisableTiming));
            h->overlap_frames = std::getenv("RT_NO_FRAME_OVERLAP") == nullptr;
            // the library's set-up per device: the address of the file-scope scratch array that
            // rt_render_frames_static works in (each device holds its own copy).  Looked up here, outside
            // any stream capture; a failure only matters to that entry point, which looks again and reports it.
            void* scratch = nullptr;
            size_t scratch_bytes = 0;
            if (rt::static_scratch(device_id, &scratch, &scratch_bytes) != hipSuccess) {
                (void)hipGetLastError();
            }
            int r = install_scene(h, std::move(scene));
            if (r != RT_OK) {
                return r;
            }
            Images images;
            r = allocate_images(h, width, height, images);
            if (r != RT_OK) {
                return r;
            }
            h->accum = images.accum;
            h->output = images.output;
            h->rng_state = images.rng_state;
            return RT_OK;
        };
        rc = body();
    }
    if (rc != RT_OK) {
        g_create_error = h->error;
        rt_destroy(h);
        return rc;
    }
    *out = h;
    return RT_OK;
}

int rt_resize(rt_handle h, uint32_t width, uint32_t height)
{
    if (h == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    if (!valid_extent(width, height)) {
        return fail_arg(h, "extent must be within 1..65536");
    }
    if (h->device < 0) {
        h->width = width;
        h->height = height;
        h->reset_accum = true;
        return RT_OK;
    }
    DeviceScope scope(h->device);
    RT_HIP_TRY(h, scope.status());
    // the new images first: if they do not fit, the handle keeps its old extent and images
    Images images;
    const int rc = allocate_images(h, width, height, images);
    if (rc != RT_OK) {
        return rc;
    }
    const hipError_t idle = hipDeviceSynchronize(); // vkDeviceWaitIdle, :203
    if (idle != hipSuccess) {
        images.release();
        return fail_hip(h, idle, "hipDeviceSynchronize()", __FILE__, __LINE__);
    }
    free_images(h); // also the per-frame sums, whose size follows the extent
    h->accum = images.accum;
    h->output = images.output;
    h->rng_state = images.rng_state;
    h->width = width;
    h->height = height;
    h->reset_accum = true; // :228
    return RT_OK;
}

int rt_destroy(rt_handle h)
{
    if (h == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    if (h->device >= 0) {
        DeviceScope scope(h->device);
        (void)hipDeviceSynchronize(); // vkDeviceWaitIdle, :238
        release_overlap(h);
        free_images(h);
        if (h->d_spheres) { (void)hipFree(h->d_spheres); }
        if (h->d_tree_geom) { (void)hipFree(h->d_tree_geom); }
        if (h->d_tree_word) { (void)hipFree(h->d_tree_word); }
        if (h->d_unit_counter) { (void)hipFree(h->d_unit_counter); }
        if (h->d_loop_geom) { (void)hipFree(h->d_loop_geom); }
        if (h->d_quad_geom) { (void)hipFree(h->d_quad_geom); }
        if (h->d_slot_spheres) { (void)hipFree(h->d_slot_spheres); }
        for (uint32_t i = 0; i < rt_tracer::kTimedLaunches; ++i) {
            if (h->ev_start[i]) { (void)hipEventDestroy(h->ev_start[i]); }
            if (h->ev_stop[i]) { (void)hipEventDestroy(h->ev_stop[i]); }
        }
        for (uint32_t i = 0; i < rt_tracer::kPipeSlots; ++i) {
            if (h->pipe_traced[i]) { (void)hipEventDestroy(h->pipe_traced[i]); }
            if (h->pipe_free[i]) { (void)hipEventDestroy(h->pipe_free[i]); }
            if (h->pipe_stream[i]) { (void)hipStreamDestroy(h->pipe_stream[i]); }
        }
        if (h->last_render) { (void)hipEventDestroy(h->last_render); }
        (void)hipGetLastError();
    }
    delete h;
    return RT_OK;
}

int rt_set_camera(rt_handle h, const float pos[3], const float dir[3], float focus_dist)
{
    if (h == nullptr || pos == nullptr || dir == nullptr) {
        return h ? fail_arg(h, "pos/dir is NULL") : RT_ERR_BAD_ARGUMENT;
    }
    h->cam_pos[0] = pos[0]; h->cam_pos[1] = pos[1]; h->cam_pos[2] = pos[2];
    const V3 nd = normalize3(v3(dir[0], dir[1], dir[2]));
    h->cam_dir[0] = nd.x; h->cam_dir[1] = nd.y; h->cam_dir[2] = nd.z;
    if (focus_dist > 0.0f) {
        h->focus_distance = focus_dist;
    }
    h->reset_accum = true;
    return RT_OK;
}

int rt_set_samples_per_pixel(rt_handle h, uint32_t spp)
{
    if (h == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    h->samples_per_pixel = std::max(1u, spp);
    h->reset_accum = true;
    return RT_OK;
}

int rt_set_aperture(rt_handle h, float aperture)
{
    if (h == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    h->aperture = std::max(0.0f, aperture);
    h->reset_accum = true;
    return RT_OK;
}

int rt_set_focus_distance(rt_handle h, float focus_dist)
{
    if (h == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    if (focus_dist > 0.0f) {
        h->focus_distance = focus_dist;
        h->reset_accum = true;
    }
    return RT_OK;
}

int rt_set_fov(rt_handle h, float vfov_deg)
{
    if (h == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    h->vertical_fov = std::clamp(vfov_deg, 5.0f, 120.0f);
    h->reset_accum = true;
    return RT_OK;
}

int rt_set_max_depth(rt_handle h, uint32_t depth)
{
    if (h == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    h->max_depth = std::max(1u, depth);
    h->reset_accum = true;
    return RT_OK;
}

int rt_set_spheres(rt_handle h, const rt_gpu_sphere* spheres, uint32_t n)
{
    if (h == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    if (spheres == nullptr || n == 0) {
        return fail_arg(h, "sphere list is empty");
    }
    if (n > rt::kMaxSpheres) {
        return fail_arg(h, "more than 1048576 spheres");
    }
    std::vector<rt_gpu_sphere> scene(spheres, spheres + n);
    int rc;
    if (h->device < 0) {
        rc = install_scene(h, std::move(scene));
    } else {
        DeviceScope scope(h->device);
        RT_HIP_TRY(h, scope.status());
        rc = install_scene(h, std::move(scene));
    }
    if (rc == RT_OK) {
        h->reset_accum = true;
    }
    return rc;
}

int rt_get_spheres(rt_handle h, rt_gpu_sphere* out, uint32_t cap, uint32_t* n)
{
    if (h == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    const uint32_t count = static_cast<uint32_t>(h->spheres.size());
    if (n != nullptr) {
        *n = count;
    }
    if (out != nullptr) {
        std::memcpy(out, h->spheres.data(), sizeof(rt_gpu_sphere) * std::min(cap, count));
    }
    return RT_OK;
}

int rt_set_scene_hierarchy(rt_handle h, int enable)
{
    if (h == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    if (enable == 2) {
        // the older tree is built and uploaded when it is first asked for
        const int rc = ensure_depth_first_tree(h, true);
        if (rc != RT_OK) {
            return rc;
        }
    }
    h->use_tree = enable != 0;
    h->use_quad = enable != 2;
    return RT_OK;
}

int rt_set_frame_overlap(rt_handle h, int enable)
{
    if (h == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    h->overlap_frames = enable != 0;
    if (!h->overlap_frames && h->device >= 0) {
        // another handle may take over once this one's overlapped launches have all been folded
        DeviceScope scope(h->device);
        bool idle = scope.status() == hipSuccess;
        for (uint32_t p = 0; p < rt_tracer::kPipeSlots && idle; ++p) {
            if (h->pipe_used[p] && hipEventQuery(h->pipe_free[p]) != hipSuccess) {
                (void)hipGetLastError();
                idle = false;
            }
        }
        if (idle) {
            release_overlap(h);
        }
    }
    return RT_OK;
}

int rt_get_scene_hierarchy(rt_handle h, float* geom, uint32_t* word, uint32_t cap, uint32_t* n_nodes)
{
    if (h == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    const int rc = ensure_depth_first_tree(h, false);
    if (rc != RT_OK) {
        return rc;
    }
    const uint32_t count = h->tree.node_count();
    if (n_nodes != nullptr) {
        *n_nodes = count;
    }
    const uint32_t take = std::min(cap, count);
    if (geom != nullptr && take != 0) {
        std::memcpy(geom, h->tree.geom.data(), sizeof(float) * 4 * take);
    }
    if (word != nullptr && take != 0) {
        if (!h->tree.packed()) {
            return fail_arg(h, "the hierarchy is too large for packed node words; use rt_get_scene_hierarchy_links");
        }
        std::memcpy(word, h->tree.word.data(), sizeof(uint32_t) * take);
    }
    return RT_OK;
}

int rt_get_scene_hierarchy_links(rt_handle h, float* geom, uint32_t* link, uint32_t* sphere, uint32_t cap, uint32_t* n_nodes)
{
    if (h == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    const int rc = ensure_depth_first_tree(h, false);
    if (rc != RT_OK) {
        return rc;
    }
    const uint32_t count = h->tree.node_count();
    if (n_nodes != nullptr) {
        *n_nodes = count;
    }
    const uint32_t take = std::min(cap, count);
    if (geom != nullptr && take != 0) {
        std::memcpy(geom, h->tree.geom.data(), sizeof(float) * 4 * take);
    }
    if (link != nullptr && take != 0) {
        std::memcpy(link, h->tree.link.data(), sizeof(uint32_t) * take);
    }
    if (sphere != nullptr && take != 0) {
        std::memcpy(sphere, h->tree.sphere.data(), sizeof(uint32_t) * take);
    }
    return RT_OK;
}

int rt_get_scene_quadtree(rt_handle h, float* box, float* geom, uint32_t* slot_sphere, uint32_t box_cap, uint32_t group_cap,
                          uint32_t slot_cap, uint32_t* n_groups, uint32_t* n_slots, uint32_t* levels, uint32_t* roots,
                          uint32_t* first_leaf, uint32_t* prologue_spheres, float* bound)
{
    if (h == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    const rt::QuadTree& q = h->quad;
    if (n_groups) { *n_groups = q.group_count(); }
    if (n_slots) { *n_slots = q.slot_count(); }
    if (levels) { *levels = q.levels; }
    if (roots) { *roots = q.roots; }
    if (first_leaf) { *first_leaf = q.first_leaf; }
    if (prologue_spheres) { *prologue_spheres = q.prologue_spheres; }
    if (bound != nullptr) {
        std::memcpy(bound, q.bound_centre, sizeof q.bound_centre);
        bound[3] = q.bound_radius;
        bound[4] = q.min_radius;
    }
    const uint32_t boxes = std::min(box_cap, q.first_leaf);
    if (box != nullptr && boxes != 0 && !q.empty()) {
        std::memcpy(box, q.box.data(), sizeof(float) * 24u * boxes);
    }
    const uint32_t groups = std::min(group_cap, static_cast<uint32_t>(q.geom.size() / 16u));
    if (geom != nullptr && groups != 0) {
        std::memcpy(geom, q.geom.data(), sizeof(float) * 16u * groups);
    }
    const uint32_t slots = std::min(slot_cap, q.slot_count());
    if (slot_sphere != nullptr && slots != 0) {
        std::memcpy(slot_sphere, q.slot_sphere.data(), sizeof(uint32_t) * slots);
    }
    return RT_OK;
}

int rt_get_params(rt_handle h, uint32_t frame_index, rt_gpu_params* out)
{
    if (h == nullptr || out == nullptr) {
        return h ? fail_arg(h, "out is NULL") : RT_ERR_BAD_ARGUMENT;
    }
    update_params(*h, frame_index, *out);
    return RT_OK;
}

int rt_reset_accumulation(rt_handle h)
{
    if (h == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    h->reset_accum = true;
    return RT_OK;
}

int rt_render(rt_handle h, uint32_t frame_index, void* stream)
{
    if (h == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    return render_impl(h, frame_index, 1, 1, band(0, h->height), true, static_cast<hipStream_t>(stream));
}

int rt_render_frames(rt_handle h, uint32_t first_frame, uint32_t n_frames, void* stream)
{
    if (h == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    return render_impl(h, first_frame, n_frames, 1, band(0, h->height), true, static_cast<hipStream_t>(stream));
}

int rt_render_shard(rt_handle h, uint32_t first_frame, uint32_t n_frames, uint32_t frame_stride,
                    uint32_t row_begin, uint32_t row_end, int write_output, void* stream)
{
    if (h == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    return render_impl(h, first_frame, n_frames, frame_stride, band(row_begin, row_end), write_output != 0,
                       static_cast<hipStream_t>(stream));
}

int rt_render_stripes(rt_handle h, uint32_t first_frame, uint32_t n_frames, uint32_t frame_stride,
                      uint32_t stripe_rows, uint32_t stripe_first, uint32_t stripe_stride, int write_output, void* stream)
{
    if (h == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    if (stripe_rows == 0 || stripe_stride == 0) {
        return fail_arg(h, "stripe_rows and stripe_stride must be >= 1");
    }
    // a stripe taller than the image is the whole image; the clamp keeps stripe_rows * width, which the
    // kernels form in 32 bits, below width * height
    stripe_rows = std::min(stripe_rows, h->height);
    rt::RowSet rows;
    rows.stripe_rows = stripe_rows;
    rows.stripe_first = stripe_first;
    rows.stripe_stride = stripe_stride;
    const uint64_t stripes_in_image = (static_cast<uint64_t>(h->height) + stripe_rows - 1) / stripe_rows;
    rows.n_stripes = stripe_first < stripes_in_image
                         ? static_cast<uint32_t>((stripes_in_image - stripe_first + stripe_stride - 1) / stripe_stride)
                         : 0u;
    if (rows.n_stripes == 0) {
        return RT_OK; // more shards than stripes: nothing to render for this one
    }
    return render_impl(h, first_frame, n_frames, frame_stride, rows, write_output != 0, static_cast<hipStream_t>(stream));
}

int rt_render_frames_static(rt_handle h, uint32_t first_frame, uint32_t n_frames, void* stream)
{
    if (h == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    return render_impl(h, first_frame, n_frames, 1, band(0, h->height), true, static_cast<hipStream_t>(stream), true);
}

size_t rt_static_scratch_bytes(void)
{
    return rt::kStaticScratchBytes - rt::kStaticScratchHeader;
}

namespace {
int resolve_impl(rt_tracer* h, const float4* accum, uint32_t sample_count, uint32_t row_begin, uint32_t row_end, hipStream_t stream)
{
    RT_NEED_DEVICE(h);
    if (row_begin > row_end || row_end > h->height) {
        return fail_arg(h, "row range outside the image");
    }
    DeviceScope scope(h->device);
    RT_HIP_TRY(h, scope.status());
    const uint64_t first = static_cast<uint64_t>(row_begin) * h->width;
    const uint64_t count = static_cast<uint64_t>(row_end - row_begin) * h->width;
    if (first + count > 0xFFFFFFFFull) {
        return fail_arg(h, "image too large for the resolve kernel");
    }
    RT_HIP_TRY(h, rt::launch_resolve(accum, h->output, static_cast<uint32_t>(first), static_cast<uint32_t>(count),
                                     static_cast<float>(sample_count), stream));
    return RT_OK;
}
} // namespace

int rt_resolve_output(rt_handle h, uint32_t sample_count, uint32_t row_begin, uint32_t row_end, void* stream)
{
    if (h == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    return resolve_impl(h, h->accum, sample_count, row_begin, row_end, static_cast<hipStream_t>(stream));
}

int rt_resolve_output_from(rt_handle h, const void* accum_device, uint32_t sample_count, uint32_t row_begin,
                           uint32_t row_end, void* stream)
{
    if (h == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    if (accum_device == nullptr) {
        return fail_arg(h, "accum_device is NULL");
    }
    return resolve_impl(h, static_cast<const float4*>(accum_device), sample_count, row_begin, row_end,
                        static_cast<hipStream_t>(stream));
}

void* rt_accum_device_ptr(rt_handle h) { return h ? h->accum : nullptr; }
void* rt_output_device_ptr(rt_handle h) { return h ? h->output : nullptr; }

int rt_read_accum(rt_handle h, float* host, size_t bytes, void* stream)
{
    if (h == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    RT_NEED_DEVICE(h);
    if (host == nullptr || bytes != static_cast<size_t>(h->width) * h->height * sizeof(float4)) {
        return fail_arg(h, "accumulator copy needs width*height*16 bytes");
    }
    DeviceScope scope(h->device);
    RT_HIP_TRY(h, scope.status());
    RT_HIP_TRY(h, hipMemcpyAsync(host, h->accum, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
    RT_HIP_TRY(h, hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return RT_OK;
}

int rt_write_accum(rt_handle h, const float* host, size_t bytes, void* stream)
{
    if (h == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    RT_NEED_DEVICE(h);
    if (host == nullptr || bytes != static_cast<size_t>(h->width) * h->height * sizeof(float4)) {
        return fail_arg(h, "accumulator copy needs width*height*16 bytes");
    }
    DeviceScope scope(h->device);
    RT_HIP_TRY(h, scope.status());
    RT_HIP_TRY(h, hipMemcpyAsync(h->accum, host, bytes, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
    RT_HIP_TRY(h, hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return RT_OK;
}

int rt_read_output(rt_handle h, uint8_t* host, size_t bytes, void* stream)
{
    if (h == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    RT_NEED_DEVICE(h);
    if (host == nullptr || bytes != static_cast<size_t>(h->width) * h->height * sizeof(uint32_t)) {
        return fail_arg(h, "output copy needs width*height*4 bytes");
    }
    DeviceScope scope(h->device);
    RT_HIP_TRY(h, scope.status());
    RT_HIP_TRY(h, hipMemcpyAsync(host, h->output, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
    RT_HIP_TRY(h, hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return RT_OK;
}

int rt_enable_rng_capture(rt_handle h, int enable)
{
    if (h == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    RT_NEED_DEVICE(h);
    DeviceScope scope(h->device);
    RT_HIP_TRY(h, scope.status());
    if (enable && h->rng_state == nullptr) {
        const size_t pixels = static_cast<size_t>(h->width) * h->height;
        RT_HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&h->rng_state), pixels * sizeof(uint32_t)));
        RT_HIP_TRY(h, hipMemset(h->rng_state, 0, pixels * sizeof(uint32_t)));
    }
    h->rng_capture = enable != 0;
    return RT_OK;
}

int rt_read_rng_state(rt_handle h, uint32_t* host, size_t bytes, void* stream)
{
    if (h == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    RT_NEED_DEVICE(h);
    if (h->rng_state == nullptr) {
        return fail_arg(h, "RNG capture was never enabled");
    }
    if (host == nullptr || bytes != static_cast<size_t>(h->width) * h->height * sizeof(uint32_t)) {
        return fail_arg(h, "RNG state copy needs width*height*4 bytes");
    }
    DeviceScope scope(h->device);
    RT_HIP_TRY(h, scope.status());
    RT_HIP_TRY(h, hipMemcpyAsync(host, h->rng_state, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
    RT_HIP_TRY(h, hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return RT_OK;
}

int rt_synchronize(rt_handle h, void* stream)
{
    if (h == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    RT_NEED_DEVICE(h);
    DeviceScope scope(h->device);
    RT_HIP_TRY(h, scope.status());
    RT_HIP_TRY(h, hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return RT_OK;
}

int rt_get_extent(rt_handle h, uint32_t* width, uint32_t* height)
{
    if (h == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    if (width) { *width = h->width; }
    if (height) { *height = h->height; }
    return RT_OK;
}

float rt_kernel_ms(rt_handle h, uint32_t launches_ago)
{
    if (h == nullptr || h->device < 0 || launches_ago >= h->launches || launches_ago >= rt_tracer::kTimedLaunches) {
        return -1.0f;
    }
    const uint32_t slot = static_cast<uint32_t>((h->launches - 1 - launches_ago) % rt_tracer::kTimedLaunches);
    DeviceScope scope(h->device);
    if (hipEventSynchronize(h->ev_stop[slot]) != hipSuccess) {
        (void)hipGetLastError();
        return -1.0f;
    }
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, h->ev_start[slot], h->ev_stop[slot]) != hipSuccess) {
        (void)hipGetLastError();
        return -1.0f;
    }
    return ms;
}

float rt_last_kernel_ms(rt_handle h)
{
    return rt_kernel_ms(h, 0);
}

#ifdef RT_STATS
// Diagnostic builds only: per-section (wave executions, active lanes) pairs accumulated since the
// handle was created; out receives 2 * rt_debug_stats_sections() 64-bit counters.
int rt_debug_stats_sections(void) { return rt::stats_section_count(); }
int rt_debug_stats_phases(void) { return rt::stats_phase_count(); }
int rt_debug_read_stats(rt_handle h, unsigned long long* out, int reset)
{
    if (h == nullptr || h->device < 0 || out == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    DeviceScope scope(h->device);
    const size_t bytes = sizeof(unsigned long long) * (2 * rt::stats_section_count() + rt::stats_phase_count());
    if (h->stats == nullptr) {
        RT_HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&h->stats), bytes));
        RT_HIP_TRY(h, hipMemset(h->stats, 0, bytes));
    }
    RT_HIP_TRY(h, hipDeviceSynchronize());
    RT_HIP_TRY(h, hipMemcpy(out, h->stats, bytes, hipMemcpyDeviceToHost));
    if (reset) {
        RT_HIP_TRY(h, hipMemset(h->stats, 0, bytes));
    }
    return RT_OK;
}
#endif

void rt_debug_fail_allocation(int nth)
{
    g_fail_allocation_in.store(nth > 0 ? nth : 0);
}

const char* rt_last_error(rt_handle h)
{
    if (h == nullptr) {
        return g_create_error.c_str();
    }
    return h->error.c_str();
}

} // extern "C"

// ---- device-side probes of single shader functions (vector tests) ----
namespace {

template <typename Body>
int with_device(int device_id, Body body)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        g_create_error = "no HIP device available";
        return RT_ERR_NO_DEVICE;
    }
    if (device_id < 0 || device_id >= count) {
        return fail_arg(nullptr, "device_id out of range");
    }
    DeviceScope scope(device_id);
    if (scope.status() != hipSuccess) {
        return fail_hip(nullptr, scope.status(), "hipSetDevice", __FILE__, __LINE__);
    }
    return body();
}

struct DeviceBuffer {
    void* ptr = nullptr;
    ~DeviceBuffer() { if (ptr) { (void)hipFree(ptr); } }
    hipError_t alloc(size_t bytes) { return hipMalloc(&ptr, bytes ? bytes : 1); }
};

} // namespace

extern "C" {

int rt_probe_hit_sphere(int device_id, const rt_gpu_sphere* spheres, const float* rays, uint32_t n, float* out, int32_t* front)
{
    if (!spheres || !rays || !out || !front || n == 0) {
        return fail_arg(nullptr, "probe arguments");
    }
    return with_device(device_id, [&]() -> int {
        DeviceBuffer ds, dr, dout, df;
        rt_tracer* none = nullptr;
        RT_HIP_TRY(none, ds.alloc(n * sizeof(rt_gpu_sphere)));
        RT_HIP_TRY(none, dr.alloc(n * 8 * sizeof(float)));
        RT_HIP_TRY(none, dout.alloc(n * 8 * sizeof(float)));
        RT_HIP_TRY(none, df.alloc(n * sizeof(int32_t)));
        RT_HIP_TRY(none, hipMemcpy(ds.ptr, spheres, n * sizeof(rt_gpu_sphere), hipMemcpyHostToDevice));
        RT_HIP_TRY(none, hipMemcpy(dr.ptr, rays, n * 8 * sizeof(float), hipMemcpyHostToDevice));
        RT_HIP_TRY(none, rt::launch_probe_hit(static_cast<rt_gpu_sphere*>(ds.ptr), static_cast<float*>(dr.ptr), n,
                                              static_cast<float*>(dout.ptr), static_cast<int32_t*>(df.ptr), nullptr));
        RT_HIP_TRY(none, hipMemcpy(out, dout.ptr, n * 8 * sizeof(float), hipMemcpyDeviceToHost));
        RT_HIP_TRY(none, hipMemcpy(front, df.ptr, n * sizeof(int32_t), hipMemcpyDeviceToHost));
        return RT_OK;
    });
}

int rt_probe_pow(int device_id, const float* x, const float* y, uint32_t n, float* out)
{
    if (!x || !y || !out || n == 0) {
        return fail_arg(nullptr, "probe arguments");
    }
    return with_device(device_id, [&]() -> int {
        DeviceBuffer dx, dy, dout;
        rt_tracer* none = nullptr;
        RT_HIP_TRY(none, dx.alloc(n * sizeof(float)));
        RT_HIP_TRY(none, dy.alloc(n * sizeof(float)));
        RT_HIP_TRY(none, dout.alloc(n * sizeof(float)));
        RT_HIP_TRY(none, hipMemcpy(dx.ptr, x, n * sizeof(float), hipMemcpyHostToDevice));
        RT_HIP_TRY(none, hipMemcpy(dy.ptr, y, n * sizeof(float), hipMemcpyHostToDevice));
        RT_HIP_TRY(none, rt::launch_probe_pow(static_cast<float*>(dx.ptr), static_cast<float*>(dy.ptr), n,
                                              static_cast<float*>(dout.ptr), nullptr));
        RT_HIP_TRY(none, hipMemcpy(out, dout.ptr, n * sizeof(float), hipMemcpyDeviceToHost));
        return RT_OK;
    });
}

int rt_probe_arith(int device_id, uint32_t op, const float* a, const float* b, uint32_t n, float* out_cheap, float* out_full)
{
    if (!a || !b || !out_cheap || !out_full || n == 0 || op > 2) {
        return fail_arg(nullptr, "probe arguments");
    }
    return with_device(device_id, [&]() -> int {
        DeviceBuffer da, db, dout;
        rt_tracer* none = nullptr;
        const size_t bytes = static_cast<size_t>(n) * sizeof(float);
        RT_HIP_TRY(none, da.alloc(bytes));
        RT_HIP_TRY(none, db.alloc(bytes));
        RT_HIP_TRY(none, dout.alloc(2 * bytes));
        RT_HIP_TRY(none, hipMemcpy(da.ptr, a, bytes, hipMemcpyHostToDevice));
        RT_HIP_TRY(none, hipMemcpy(db.ptr, b, bytes, hipMemcpyHostToDevice));
        RT_HIP_TRY(none, rt::launch_probe_arith(op, static_cast<float*>(da.ptr), static_cast<float*>(db.ptr), n,
                                                static_cast<float*>(dout.ptr), nullptr));
        RT_HIP_TRY(none, hipMemcpy(out_cheap, dout.ptr, bytes, hipMemcpyDeviceToHost));
        RT_HIP_TRY(none, hipMemcpy(out_full, static_cast<float*>(dout.ptr) + n, bytes, hipMemcpyDeviceToHost));
        return RT_OK;
    });
}

int rt_probe_rng(int device_id, const uint32_t* xyz, uint32_t n, uint32_t draws, uint32_t* out)
{
    if (!xyz || !out || n == 0) {
        return fail_arg(nullptr, "probe arguments");
    }
    return with_device(device_id, [&]() -> int {
        DeviceBuffer din, dout;
        rt_tracer* none = nullptr;
        const size_t out_bytes = static_cast<size_t>(n) * (1 + draws) * sizeof(uint32_t);
        RT_HIP_TRY(none, din.alloc(n * 3 * sizeof(uint32_t)));
        RT_HIP_TRY(none, dout.alloc(out_bytes));
        RT_HIP_TRY(none, hipMemcpy(din.ptr, xyz, n * 3 * sizeof(uint32_t), hipMemcpyHostToDevice));
        RT_HIP_TRY(none, rt::launch_probe_rng(static_cast<uint32_t*>(din.ptr), n, draws, static_cast<uint32_t*>(dout.ptr), nullptr));
        RT_HIP_TRY(none, hipMemcpy(out, dout.ptr, out_bytes, hipMemcpyDeviceToHost));
        return RT_OK;
    });
}

} // extern "C"

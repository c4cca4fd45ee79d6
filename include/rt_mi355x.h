/*
 * rt_mi355x.h -- C ABI of the MI355X (gfx950) sphere path tracer.
 *
 * This library replaces the traced path of the reference renderer: the class
 * RayTracer (src/rt/RayTracer.h:34-94), its Vulkan dispatch
 * (src/rt/RayTracer.cpp:559-685) and the compute shader it launches
 * (shaders/raytrace.comp.glsl).  Every entry point below names the reference
 * interface it stands in for.  All functions return 0 on success or a
 * non-zero rt_status; nothing throws across this boundary.  A handle is used
 * from one thread at a time and from one stream at a time (a call on another
 * stream must be ordered after the handle's earlier work by the caller: the
 * launches of one handle share its accumulator, its per-frame sums and its unit
 * counter); one handle drives one GPU.  The one exception: rt_resolve_output_from
 * reads a caller-owned image and writes the output image only, so it may run on
 * another stream beside render calls of the same handle that pass
 * write_output = 0 (a multi-GPU step resolves the gathered image of step i while
 * step i + 1 renders).  A call that fails leaves the handle's extent, images and
 * scene as they were before the call.
 *
 * Images are linear row-major, row 0 = top, pitch = width: the accumulator is
 * float4 per pixel (the rgba32f storage image, binding 0), the output is
 * uchar4 per pixel (the rgba8 swapchain image, binding 1).
 */
#ifndef RT_MI355X_H
#define RT_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* GPUSphere, src/rt/RayTracer.h:12-17 == struct Sphere, raytrace.comp.glsl:7-12
 * (std430, stride 48). */
typedef struct rt_gpu_sphere {
    float center_radius[4]; /* xyz centre, w radius */
    float albedo[4];        /* rgb albedo, w unused */
    float misc[4];          /* x material (0 lambert, 1 metal, 2 dielectric), y fuzz, z refIdx, w flags (>0.5: checker) */
} rt_gpu_sphere;

/* GPUParams, src/rt/RayTracer.h:20-32 == uniform Params, raytrace.comp.glsl:19-31
 * (std140, 160 bytes). */
typedef struct rt_gpu_params {
    float origin_lens[4];   /* xyz origin, w lens radius */
    float lower_left[4];
    float horizontal[4];
    float vertical[4];
    float u[4];
    float v[4];
    float w[4];
    uint32_t frame_sample_depth_count[4]; /* frame, samples per frame, max depth, sphere count */
    float resolution[4];    /* x width, y height */
    float inv_resolution[4];
} rt_gpu_params;

typedef struct rt_tracer *rt_handle;

/* device_id for rt_create that builds a handle holding host state only (no HIP
 * call is made): setters, rt_get_params, rt_get_spheres and rt_resize work,
 * every call that needs the GPU returns RT_ERR_NO_DEVICE. */
#define RT_DEVICE_NONE (-1)

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_BAD_ARGUMENT = 1,
    RT_ERR_HIP = 2,          /* a HIP runtime call failed; see rt_last_error */
    RT_ERR_NO_DEVICE = 3,
    RT_ERR_OUT_OF_MEMORY = 4
} rt_status;

/* RayTracer::create, src/rt/RayTracer.cpp:179-199.  Allocates the accumulator
 * and output images for width x height on HIP device device_id, installs the
 * default five-sphere scene (:105-138), aims the default camera (13,2,3) at
 * (0,1,0) with focus = distance (:189-193) and raises the reset flag.  Member
 * defaults are those of RayTracer.h:87-93 (spp 4, depth 12, aperture 0.05,
 * vfov 20). */
int rt_create(int device_id, uint32_t width, uint32_t height, rt_handle *out);

/* RayTracer::resize, src/rt/RayTracer.cpp:201-234: waits for the device,
 * re-allocates the images at the new extent, raises the reset flag. */
int rt_resize(rt_handle h, uint32_t width, uint32_t height);

/* RayTracer::destroy, src/rt/RayTracer.cpp:236-296. */
int rt_destroy(rt_handle h);

/* RayTracer::setCamera, src/rt/RayTracer.cpp:298-309: stores pos and
 * normalize(dir); focus_dist is taken only when > 0; raises the reset flag. */
int rt_set_camera(rt_handle h, const float pos[3], const float dir[3], float focus_dist);
/* RayTracer::setSamplesPerPixel, :311-315 (max(1, spp)). */
int rt_set_samples_per_pixel(rt_handle h, uint32_t spp);
/* RayTracer::setAperture, :317-321 (max(0, aperture)). */
int rt_set_aperture(rt_handle h, float aperture);
/* RayTracer::setFocusDistance, :323-330 (ignored, flag untouched, unless > 0). */
int rt_set_focus_distance(rt_handle h, float focus_dist);
/* RayTracer::setFov, :332-336 (clamped to [5, 120] degrees). */
int rt_set_fov(rt_handle h, float vfov_deg);
/* RayTracer::setMaxDepth, :338-342 (max(1, depth)). */
int rt_set_max_depth(rt_handle h, uint32_t depth);

/* Replaces the sphere list that RayTracer::buildScene/uploadScene
 * (src/rt/RayTracer.cpp:105-138, :529-548) hard-code.  The reference has no
 * public counterpart; this is what larger scenes enter through.  1 <= n <= 2^20.
 * Raises the reset flag. */
int rt_set_spheres(rt_handle h, const rt_gpu_sphere *spheres, uint32_t n);
/* Copies out up to cap spheres of the current scene; *n receives the count. */
int rt_get_spheres(rt_handle h, rt_gpu_sphere *out, uint32_t cap, uint32_t *n);

/* Scenes of 24 spheres or more are traced through a hierarchy of bounding
 * volumes that rt_set_spheres builds: a ray is only tested against the spheres
 * whose enclosing volumes it can reach, each with the arithmetic of hitSphere
 * (shaders/raytrace.comp.glsl:109-133), and the closest-hit rule of :247-259 is
 * kept (smallest root, the higher index on an exact tie), so images are
 * bit-identical to testing every sphere.  The reference has no counterpart (its
 * loop visits all spheres).  enable = 0 goes back to that loop; the default is 1.
 * The hierarchy is a complete four-way tree of bounding boxes, four children per
 * step (rt_get_scene_quadtree), kept in the GPU's on-chip memory for scenes of up
 * to about 6000 spheres and in global memory beyond; enable = 2 walks the
 * depth-first tree of balls below instead, one node per step (same images; for
 * comparisons).
 * Does not raise the reset flag. */
int rt_set_scene_hierarchy(rt_handle h, int enable);
/* Copies out up to cap nodes of the hierarchy of the current scene, in
 * depth-first order: geom receives 4 floats per node (centre; radius^2 for a
 * leaf, 1.125 R^2 for an inner node), word one 32-bit value per node (bit 31
 * leaf, bits 17..30 sphere index of a leaf, bits 0..16 byte offset = 16 * index
 * of the node to continue with when the node's test fails).  *n_nodes receives
 * the node count, 0 when the scene has no hierarchy.  Either array may be NULL.
 * A hierarchy of more than 7850 nodes is not kept in this format: asking for
 * its words fails (geom and *n_nodes are still delivered); use the call below. */
int rt_get_scene_hierarchy(rt_handle h, float *geom, uint32_t *word, uint32_t cap, uint32_t *n_nodes);
/* The same hierarchy with whole words, for trees of any size (scenes of up to
 * 2^20 spheres): link receives per node the index of the node to continue with
 * when the node's test fails, sphere the sphere index of a leaf or 0xFFFFFFFF for
 * an inner node.  No counterpart in the reference.  Any array may be NULL. */
int rt_get_scene_hierarchy_links(rt_handle h, float *geom, uint32_t *link, uint32_t *sphere, uint32_t cap,
                                 uint32_t *n_nodes);

/* The hierarchy of the current scene as the complete four-way tree that is
 * walked (on chip when it fits, in global memory otherwise).  The tree has
 * *levels levels; level 0 has *roots groups, every level below four times as many; the children of group g are the
 * groups 4 g + *roots + k, k = 0..3.  Groups before *first_leaf hold the
 * axis-aligned bounding boxes of their four child groups: box receives 24 floats
 * per such group, lo.x[4], lo.y[4], lo.z[4], hi.x[4], hi.y[4], hi.z[4] (an unused
 * child: 3e38 everywhere).  Groups from *first_leaf on hold four spheres each:
 * geom receives 16 floats per such group, (centre, radius^2) four times, and
 * slot_sphere[4 (g - *first_leaf) + k] is the index of the sphere in slot k of
 * group g; an unused slot has radius^2 = -infinity and the index 0xFFFFFFFF.  The
 * last group (index *n_groups - 1, behind the tree) holds the *prologue_spheres
 * (0..4) spheres that are far larger than the rest and are tested for every ray.
 * bound receives five floats: centre and radius of a ball around all spheres of
 * the tree, and their smallest radius (the walk's per-ray box margin is derived
 * from them).  *n_groups = 0: the scene has no such tree.  No counterpart in the
 * reference.  Any pointer may be NULL. */
int rt_get_scene_quadtree(rt_handle h, float *box, float *geom, uint32_t *slot_sphere, uint32_t box_cap,
                          uint32_t group_cap, uint32_t slot_cap, uint32_t *n_groups, uint32_t *n_slots,
                          uint32_t *levels, uint32_t *roots, uint32_t *first_leaf, uint32_t *prologue_spheres,
                          float *bound);

/* RayTracer::makeCameraParams + updateParams, src/rt/RayTracer.cpp:140-177,
 * :550-557: the 160-byte block a dispatch for frame_index would receive. */
int rt_get_params(rt_handle h, uint32_t frame_index, rt_gpu_params *out);

/* Raises the reset flag without changing any setting (what every setter and
 * create/resize do as a side effect, RayTracer.h:81). */
int rt_reset_accumulation(rt_handle h);

/* RayTracer::render, src/rt/RayTracer.cpp:559-685: clears the accumulator iff
 * the reset flag is up or frame_index == 0 (:564), lowers the flag, then runs
 * one dispatch of the shader for frame_index (raytrace.comp.glsl:298-343) over
 * the whole image.  The GPU work is asynchronous on `stream` (a hipStream_t;
 * NULL = the device's default stream), like the reference's recording into the
 * caller's command buffer, and successive calls on one stream are ordered.  The
 * call itself may block on the host for a while: with launch overlap on (the
 * default, see rt_set_frame_overlap) it waits until the call before the previous
 * one has been added to the accumulator, and the first overlapped call of a
 * handle allocates its two per-frame sum images.  On a stream that is being
 * captured into a HIP graph the call is not overlapped: everything it launches
 * goes to `stream` and becomes part of the graph. */
int rt_render(rt_handle h, uint32_t frame_index, void *stream);

/* Consecutive single-frame calls (rt_render, or a shard call with n_frames == 1)
 * overlap on the GPU by default: the library traces call k on its own stream
 * k % 2 into a per-frame sum image (two per handle, 32 bytes per pixel together,
 * allocated by the first call that overlaps) and adds the sums to the accumulator
 * on the caller's stream in call order, so
 * the end of one launch, when the GPU runs empty, is filled by the start of the
 * next.  Accumulator, output and their ordering on the caller's stream are those
 * of the plain dispatch (raytrace.comp.glsl:333-342).  enable = 0 keeps every
 * launch on the caller's stream; a call on a stream that is being captured into
 * a HIP graph does so by itself.  The reference serialises its dispatches with a semaphore
 * (src/core/App.cpp:163-168) because each one reads the accumulator; here only
 * the short fold does.  Two consequences for the caller: an overlapped call
 * may wait on the host until the call before the previous one has been folded
 * (the library's streams never wait on the device, so that no ring of waiting
 * streams can form when the runtime lets streams share a hardware queue); and
 * of several handles on one GPU only one overlaps at a time, the first that
 * makes such a call, until it is destroyed or has its overlap switched off
 * while idle (launches of different handles run beside each other anyway). */
int rt_set_frame_overlap(rt_handle h, int enable);

/* n_frames consecutive rt_render calls (frame_index = first_frame ..
 * first_frame + n_frames - 1) as one launch: same accumulator bits, the output
 * image is that of the last frame.  This is the caller loop of
 * src/core/App.cpp:403,527 folded into the device. */
int rt_render_frames(rt_handle h, uint32_t first_frame, uint32_t n_frames, void *stream);

/* The sharded form of rt_render_frames used across GPUs: renders frames
 * first_frame, first_frame + frame_stride, ... (n_frames of them) for rows
 * [row_begin, row_end) only.  Pixel coordinates stay global, so every texel
 * gets the same RNG stream as on one GPU (raytrace.comp.glsl:310-311).  Clear
 * rule: a raised reset flag clears the whole image (and is lowered);
 * first_frame == 0 clears rows [row_begin,row_end) only, so the bands of one
 * image can be rendered by successive calls.  With
 * write_output == 0 the rgba8 image is left untouched (a frame-split rank
 * resolves it after the reduction with rt_resolve_output). */
int rt_render_shard(rt_handle h, uint32_t first_frame, uint32_t n_frames, uint32_t frame_stride,
                    uint32_t row_begin, uint32_t row_end, int write_output, void *stream);

/* rt_render_shard with interleaved rows, for load balance across GPUs (the sky
 * rows of an image are cheaper than its ground rows): the image is cut into
 * stripes of stripe_rows rows, stripe k = rows [k * stripe_rows,
 * min(height, (k + 1) * stripe_rows)), and the call renders stripes
 * stripe_first, stripe_first + stripe_stride, ... to the bottom of the image
 * (rank r of G passes stripe_first = r, stripe_stride = G).  Pixel coordinates
 * stay global (raytrace.comp.glsl:310-311), so the stripes of all ranks put
 * together are the unsharded image bit for bit.  Clear rule as for
 * rt_render_shard, with "the rows this call renders".  No counterpart in the
 * reference (one device, src/vk/VulkanContext.cpp:413-414). */
int rt_render_stripes(rt_handle h, uint32_t first_frame, uint32_t n_frames, uint32_t frame_stride,
                      uint32_t stripe_rows, uint32_t stripe_first, uint32_t stripe_stride, int write_output,
                      void *stream);

/* rt_render_frames for callers that cannot allocate device memory (under HIP
 * graph capture, in a tight serving loop, in harnesses that forbid hipMalloc):
 * the per-frame sums of a multi-frame launch and the unit counter of the
 * hierarchy kernel live in a file-scope device array of
 * rt_static_scratch_bytes() bytes (64 MiB) that every GPU the library runs on
 * holds from rt_create until the process ends, and the call allocates and frees
 * nothing.  Same accumulator, RNG and output bits as rt_render_frames.  The
 * frames are traced in groups of as many as the array holds at a time; a call
 * with n_frames > 1 on an image of which two frames do not fit (more than
 * rt_static_scratch_bytes() / 32 = 2 097 152 pixels; 1920x1080 still fits) is
 * rejected with RT_ERR_BAD_ARGUMENT and a message that states both sizes.  All
 * handles of one GPU share the array: calls that overlap in time on one GPU
 * (different handles or streams) race.  It has no counterpart in the reference,
 * whose Vulkan path allocates at create/resize only
 * (src/rt/RayTracer.cpp:498-527). */
int rt_render_frames_static(rt_handle h, uint32_t first_frame, uint32_t n_frames, void *stream);
size_t rt_static_scratch_bytes(void);

/* The display tail of the shader alone, raytrace.comp.glsl:337-342: divides the
 * accumulator by float(sample_count), applies c/(c+1) and pow(max(c,0),1/2.2),
 * stores rgba8 for rows [row_begin,row_end).  sample_count is the shader's
 * (frame + 1) * spp. */
int rt_resolve_output(rt_handle h, uint32_t sample_count, uint32_t row_begin, uint32_t row_end, void *stream);
/* The same, reading a caller-owned float4 image of the handle's extent (e.g. the
 * buffer a multi-GPU sum-reduction of accumulators landed in).  Touches nothing
 * of the handle but the output image: see the stream rule at the top. */
int rt_resolve_output_from(rt_handle h, const void *accum_device, uint32_t sample_count, uint32_t row_begin,
                           uint32_t row_end, void *stream);

/* Device addresses of the two images (bindings 0 and 1 of
 * raytrace.comp.glsl:33-34).  Valid until rt_resize/rt_destroy. */
void *rt_accum_device_ptr(rt_handle h);
void *rt_output_device_ptr(rt_handle h);

/* Blocking copies (they synchronise `stream` first).  bytes must equal
 * width*height*16 (accum) or width*height*4 (output). */
int rt_read_accum(rt_handle h, float *host, size_t bytes, void *stream);
int rt_write_accum(rt_handle h, const float *host, size_t bytes, void *stream);
int rt_read_output(rt_handle h, uint8_t *host, size_t bytes, void *stream);

/* Test hook: when enabled, every dispatch also stores the RNG state each pixel
 * ends with (after its last sample of the last frame) so that the per-pixel
 * stream of raytrace.comp.glsl:310-331 can be compared word for word. */
int rt_enable_rng_capture(rt_handle h, int enable);
int rt_read_rng_state(rt_handle h, uint32_t *host, size_t bytes, void *stream);

/* Test hook: the nth device allocation the library makes from now on (1 = the
 * next one) fails with an out-of-memory error; 0 switches the hook off.  Lets
 * tests reach the error paths of rt_resize / rt_set_spheres / rt_render_frames. */
void rt_debug_fail_allocation(int nth);

/* Waits for `stream` (vkDeviceWaitIdle in the reference's resize/destroy). */
int rt_synchronize(rt_handle h, void *stream);

int rt_get_extent(rt_handle h, uint32_t *width, uint32_t *height);

/* Milliseconds the most recent trace kernel of this handle took on the GPU,
 * measured with HIP events on the stream it was launched on; blocks until that
 * kernel has finished.  Negative if nothing has been launched. */
float rt_last_kernel_ms(rt_handle h);
/* The same for the trace kernel launched `launches_ago` launches before the most
 * recent one (0 = most recent; the last 64 launches are kept). */
float rt_kernel_ms(rt_handle h, uint32_t launches_ago);

/* Text of the last error on this handle (VK_CHECK's message in the reference,
 * src/util/Check.h:8-19).  h may be NULL: then the last rt_create failure. */
const char *rt_last_error(rt_handle h);

/* Device-side probes of single shader functions, for vector tests: evaluate
 * the function for n inputs on the GPU with the same code the tracer uses. */
/* hitSphere, raytrace.comp.glsl:109-153.  rays: n x 8 floats (o.xyz, tmin, d.xyz, tmax);
 * spheres: n spheres; out: n x 8 floats (hit flag, t, p.xyz, n.xyz) ; front: n ints. */
int rt_probe_hit_sphere(int device_id, const rt_gpu_sphere *spheres, const float *rays, uint32_t n,
                        float *out, int32_t *front);
/* pow as the tracer evaluates it (schlick :196, tonemap :340). */
int rt_probe_pow(int device_id, const float *x, const float *y, uint32_t n, float *out);
/* The tracer's divide / sqrt / inversesqrt helpers (which skip the range-scaling steps of the IEEE
 * sequences when the operands allow) next to the full sequences, element-wise: op 0: a / b,
 * op 1: sqrt(a), op 2: 1 / sqrt(a).  Both outputs must equal the correctly rounded result. */
int rt_probe_arith(int device_id, uint32_t op, const float *a, const float *b, uint32_t n, float *out_cheap,
                   float *out_full);
/* hash/lcg/rand, raytrace.comp.glsl:37-55: for each of n seeds (x,y,z) returns the
 * hash and the first `draws` lcg states. out: n x (1 + draws) words. */
int rt_probe_rng(int device_id, const uint32_t *xyz, uint32_t n, uint32_t draws, uint32_t *out);

#ifdef __cplusplus
}
#endif
#endif /* RT_MI355X_H */

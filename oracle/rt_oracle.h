/*
 * CPU oracle for the sphere path tracer -- TEST INFRASTRUCTURE ONLY.
 *
 * This is a scalar C restatement of the reference compute shader
 * (shaders/raytrace.comp.glsl:37-343) and of the host code that feeds it
 * (src/rt/RayTracer.cpp:105-177, :550-557).  It exists so that the HIP path
 * can be checked against an independent implementation.  Nothing in the
 * shipped library links, loads or calls it: only tests/, the smoke check and
 * the cpu_baseline leg of bench.py do.
 *
 * Pinning: the reference ships no tests or golden images.  The host half is
 * pinned by golden parameter/sphere bytes produced by compiling the
 * reference's own RayTracer.cpp (oracle/_ref).  The integer RNG is pinned by
 * known-answer vectors derived from the shader text.  The GLSL built-ins the
 * shader calls (dot, normalize, inversesqrt, reflect, mix, mod, floor, min,
 * max) are pinned bit for bit by vectors computed with the reference's
 * vendored GLM 1.0.1 (oracle/ref_glm_probe.cpp, tests/golden/glm_builtins.json).
 * The shader as a whole cannot be run here (no GLSL compiler, no Vulkan ICD),
 * so for the composition of those pieces -- control flow, operand order of
 * the shader's own expressions, pow -- it remains: PARITY UNPINNED.  The
 * restatement follows the shader source order with every operation rounded
 * once to IEEE-754 binary32 (no contraction).
 *
 * Arithmetic conventions (GLSL built-ins are restated, see rt_oracle.c):
 *   dot(a,b)        = (a.x*b.x + a.y*b.y) + a.z*b.z
 *   inversesqrt(x)  = 1.0f / sqrtf(x)          (two correctly rounded steps)
 *   normalize(v)    = v * inversesqrt(dot(v,v))
 *   reflect(I,N)    = I - (2*dot(N,I))*N
 *   mix(x,y,a)      = x*(1-a) + y*a
 *   mod(x,y)        = x - y*floor(x/y)
 *   min(x,y)        = y < x ? y : x ;  max(x,y) = x < y ? y : x
 *   pow(x,y)        = exp2(y*log2(x)) with the portable float kernels
 *                     rto_log2f / rto_exp2f defined in rt_oracle.c
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/rt/RayTracer.h:12-17 (GPUSphere, 48 bytes, std430 stride 48) */
typedef struct rto_sphere {
    float center_radius[4];
    float albedo[4];
    float misc[4]; /* material, fuzz, refIdx, flags */
} rto_sphere;

/* src/rt/RayTracer.h:20-32 (GPUParams, 160 bytes, std140) */
typedef struct rto_params {
    float origin_lens[4];
    float lower_left[4];
    float horizontal[4];
    float vertical[4];
    float u[4];
    float v[4];
    float w[4];
    uint32_t frame_sample_depth_count[4];
    float resolution[4];
    float inv_resolution[4];
} rto_params;

/* Event counters for the algorithmic-FLOP model.  All counts are totals over
 * the rendered region. */
typedef struct rto_counters {
    uint64_t samples;          /* camera rays generated */
    uint64_t disk_iters;       /* randomInUnitDisk loop iterations */
    uint64_t segments;         /* closest-hit searches (bounces evaluated) */
    uint64_t sphere_tests;     /* hitSphere calls */
    uint64_t disc_nonneg;      /* tests that reached the sqrt */
    uint64_t far_roots;        /* tests that evaluated the far root */
    uint64_t accepts;          /* hitSphere returned true */
    uint64_t sky;              /* paths ended on the sky */
    uint64_t lambert;          /* scatterLambert calls */
    uint64_t lambert_checker;  /* of which took the checker branch */
    uint64_t metal;            /* scatterMetal calls */
    uint64_t metal_absorbed;   /* of which returned false */
    uint64_t dielectric;       /* scatterDielectric calls */
    uint64_t diel_schlick;     /* schlick() evaluated (cannotRefract false) */
    uint64_t diel_refract;     /* refractVec called */
    uint64_t sphere_iters;     /* randomInUnitSphere loop iterations */
    uint64_t depth_exhausted;  /* paths that ran out of bounces */
    uint64_t rand_draws;       /* rand() calls */
    uint64_t pixels;           /* pixel invocations */
    uint64_t diel_front;       /* dielectric hits on the front face (1/refIdx evaluated) */
} rto_counters;

/* raytrace.comp.glsl:37-55 */
uint32_t rto_hash3(uint32_t x, uint32_t y, uint32_t z);
uint32_t rto_lcg(uint32_t *state);
float    rto_rand(uint32_t *state);

/* portable transcendental kernels used for GLSL pow() */
float rto_log2f(float x);
float rto_exp2f(float x);
float rto_powf(float x, float y);

/* src/rt/RayTracer.cpp:105-138 */
void rto_default_scene(rto_sphere out[5]);

/* glm::normalize, as setCamera applies to the direction it stores
 * (src/rt/RayTracer.cpp:301). */
void rto_normalize3(const float in[3], float out[3]);
/* The GLSL built-ins as this restatement implements them, by name ("dot", "normalize", "reflect",
 * "cross", "length", "mix", "mod", "floor", "inversesqrt", "min", "max", "pow"): inputs in `in`
 * (vectors first, then scalars), results in `out`; returns the number of results or -1. */
int rto_builtin(const char *op, const float *in, float *out);

/* src/rt/RayTracer.cpp:140-177; dir is the stored camera direction (already
 * normalised once by create() or setCamera()); it is normalised once more
 * inside, as :143 does.  frame_sample_depth_count is left zero, as
 * makeCameraParams does. */
void rto_make_camera_params(const float pos[3], const float dir[3],
                            float aperture, float vfov_deg, float focus,
                            uint32_t width, uint32_t height, rto_params *out);

/* The create-time aim of src/rt/RayTracer.cpp:189-193: dir and focus for the
 * default camera position looking at (0,1,0). */
void rto_default_aim(const float pos[3], float dir_out[3], float *focus_out);

/*
 * One dispatch of the shader main() (raytrace.comp.glsl:298-343) over rows
 * [y0, y1) of the image.  accum is W*H float4, row 0 = top; it is
 * read-modify-written exactly as the shader does (prev + accum, alpha = 1).
 * out_rgba8 (W*H*4 bytes) and rng_final (W*H words, RNG state after the last
 * sample of each pixel) may be NULL.  counters may be NULL.  nthreads <= 0
 * means 1.  Returns 0.
 */
int rto_render(const rto_params *params, const rto_sphere *spheres,
               float *accum, uint8_t *out_rgba8, uint32_t *rng_final,
               uint32_t y0, uint32_t y1, int nthreads, rto_counters *counters);

/* The display tail of the shader alone (raytrace.comp.glsl:337-342) over `texels` float4 texels:
 * sample_count is the shader's (frame + 1) * spp, already wrapped to 32 bits. */
void rto_resolve(const float *accum, uint8_t *out_rgba8, size_t texels, uint32_t sample_count);

/* Per-function probes for vector tests (exact float comparison with HIP). */
/* in: ray o[3], d[3], sphere; out: returns hit flag, fills t,p,n,front */
int rto_hit_sphere(const rto_sphere *s, const float o[3], const float d[3],
                   float tmin, float tmax, float *t, float p[3], float n[3],
                   int *front);
/* Trace one camera sample of pixel (px,py) starting from RNG state *rng:
 * restates the body of the spp loop (raytrace.comp.glsl:317-330).  Returns
 * the radiance in rgb and advances *rng. */
void rto_sample(const rto_params *params, const rto_sphere *spheres,
                uint32_t px, uint32_t py, uint32_t *rng, float rgb[3]);

#ifdef __cplusplus
}
#endif
#endif

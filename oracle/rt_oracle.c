/*
 * CPU oracle for the sphere path tracer -- TEST INFRASTRUCTURE ONLY.
 * See rt_oracle.h for scope, pinning status and arithmetic conventions.
 *
 * Build: gcc -O2 -std=c99 -ffp-contract=off -fno-fast-math -fPIC -shared
 * Every floating-point operation below is a single IEEE-754 binary32
 * operation in shader source order; nothing may be fused or re-associated.
 */
#include "rt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* small vector helpers (component-wise, one rounding per operation)   */
/* ------------------------------------------------------------------ */
typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 divs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
static inline v3 neg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline float dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float glsl_min(float x, float y) { return (y < x) ? y : x; }
static inline float glsl_max(float x, float y) { return (x < y) ? y : x; }
static inline float inversesqrt(float x) { return 1.0f / sqrtf(x); }
static inline v3 normalize(v3 a) { return scale(a, inversesqrt(dot(a, a))); }
/* reflect(I, N) = I - 2.0 * dot(N, I) * N */
static inline v3 reflect(v3 i, v3 n) { return sub(i, scale(n, 2.0f * dot(n, i))); }
/* GLSL mix(x, y, a) = x * (1 - a) + y * a, per component with a scalar a */
static inline v3 mix3(v3 x, v3 y, float a)
{
    float oma = 1.0f - a;
    return V(x.x * oma + y.x * a, x.y * oma + y.y * a, x.z * oma + y.z * a);
}
/* GLSL mod(x, y) = x - y * floor(x / y) */
static inline float glsl_mod(float x, float y) { return x - y * floorf(x / y); }

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* ------------------------------------------------------------------ */
/* RNG: raytrace.comp.glsl:37-55                                       */
/* ------------------------------------------------------------------ */
static inline uint32_t hash1(uint32_t v)
{
    v = (v ^ (v >> 17)) * 0xED5AD4BBu;
    v = (v ^ (v >> 11)) * 0xAC4C1B51u;
    v = (v ^ (v >> 15)) * 0x31848BABu;
    return v;
}

uint32_t rto_hash3(uint32_t x, uint32_t y, uint32_t z)
{
    return hash1(x) ^ hash1(y) ^ hash1(z);
}

uint32_t rto_lcg(uint32_t *state)
{
    *state = 1664525u * (*state) + 1013904223u;
    return *state;
}

float rto_rand(uint32_t *state)
{
    return (float)(rto_lcg(state) & 0x00FFFFFFu) / (float)0x01000000u;
}

/* ------------------------------------------------------------------ */
/* pow(): GLSL leaves its precision to exp2(y*log2(x)).  These kernels */
/* use only +,-,*,/ and integer bit operations so that any IEEE-754    */
/* implementation reproduces them bit for bit.                         */
/* ------------------------------------------------------------------ */
float rto_log2f(float x)
{
    uint32_t ix = f2u(x);
    int k = 0;
    if (ix >= 0x7f800000u) {
        return (ix > 0x80000000u && ix <= 0xff800000u) ? u2f(0x7fc00000u) : x; /* x<0: NaN; inf/NaN: x */
    }
    if (ix == 0u) {
        return u2f(0xff800000u); /* -inf */
    }
    if (ix < 0x00800000u) {
        x = x * 8388608.0f;
        k = -23;
        ix = f2u(x);
    }
    k += (int)(ix >> 23) - 127;
    ix &= 0x007fffffu;
    uint32_t i = (ix + 0x004afb20u) & 0x00800000u;
    float m = u2f(ix | (i ^ 0x3f800000u));
    k += (int)(i >> 23);
    float f = m - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    float w = z * z;
    float t1 = w * (0.40000972152f + w * 0.24279078841f);
    float t2 = z * (0.66666662693f + w * 0.28498786688f);
    float R = t2 + t1;
    float hfsq = (0.5f * f) * f;
    float lnm = f - (hfsq - s * (hfsq + R));
    return (float)k + lnm * 1.44269504089f;
}

static inline float pow2i(int n) /* 2^n for n in [-126, 127] */
{
    return u2f((uint32_t)(n + 127) << 23);
}

float rto_exp2f(float p)
{
    if (!(p == p)) {
        return p;
    }
    if (p >= 128.0f) {
        return u2f(0x7f800000u);
    }
    if (p < -150.0f) {
        return 0.0f;
    }
    float n = floorf(p + 0.5f);
    float r = p - n;
    float q = 1.52527338041e-5f;
    q = 1.54035303934e-4f + r * q;
    q = 1.33335581464e-3f + r * q;
    q = 9.61812910763e-3f + r * q;
    q = 5.55041086648e-2f + r * q;
    q = 2.40226506959e-1f + r * q;
    q = 6.93147180560e-1f + r * q;
    q = 1.0f + r * q;
    int ni = (int)n;
    if (ni > 127) {
        return (q * pow2i(127)) * pow2i(ni - 127);
    }
    if (ni < -126) {
        return (q * pow2i(ni + 24)) * pow2i(-24);
    }
    return q * pow2i(ni);
}

float rto_powf(float x, float y)
{
#ifdef RTO_LIBM_POW
    /* the conforming-variant build (oracle/Makefile, target `variant`): pow as the C library rounds it, one
     * of the many precisions GLSL allows an implementation */
    return powf(x, y);
#endif
    if (x == 0.0f) {
        return (y > 0.0f) ? 0.0f : ((y == 0.0f) ? 1.0f : u2f(0x7f800000u));
    }
    return rto_exp2f(y * rto_log2f(x));
}

/* ------------------------------------------------------------------ */
/* host half: src/rt/RayTracer.cpp                                     */
/* ------------------------------------------------------------------ */
static void set_sphere(rto_sphere *s, float cx, float cy, float cz, float r,
                       float ar, float ag, float ab,
                       float mat, float fuzz, float ior, float flags)
{
    s->center_radius[0] = cx; s->center_radius[1] = cy; s->center_radius[2] = cz; s->center_radius[3] = r;
    s->albedo[0] = ar; s->albedo[1] = ag; s->albedo[2] = ab; s->albedo[3] = 0.0f;
    s->misc[0] = mat; s->misc[1] = fuzz; s->misc[2] = ior; s->misc[3] = flags;
}

/* src/rt/RayTracer.cpp:105-138 */
void rto_default_scene(rto_sphere out[5])
{
    set_sphere(&out[0], 0.0f, -1000.0f, 0.0f, 1000.0f, 0.75f, 0.8f, 0.9f, 0.0f, 0.0f, 1.0f, 1.0f);
    set_sphere(&out[1], 0.0f, 1.0f, 0.0f, 1.0f, 0.9f, 0.25f, 0.25f, 0.0f, 0.0f, 1.0f, 0.0f);
    set_sphere(&out[2], -4.0f, 1.0f, 0.0f, 1.0f, 1.0f, 1.0f, 1.0f, 2.0f, 0.0f, 1.5f, 0.0f);
    set_sphere(&out[3], 4.0f, 1.0f, 0.0f, 1.0f, 0.95f, 0.65f, 0.15f, 1.0f, 0.03f, 1.0f, 0.0f);
    set_sphere(&out[4], 2.5f, 0.5f, 2.5f, 0.5f, 0.95f, 0.95f, 0.98f, 1.0f, 0.0f, 1.0f, 0.0f);
}

/* GLM 1.0.1 semantics: normalize = v * (1/sqrt(dot)), cross as written. */
static inline v3 cross(v3 a, v3 b)
{
    return V(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}

/* src/rt/RayTracer.cpp:189-193 */
void rto_default_aim(const float pos[3], float dir_out[3], float *focus_out)
{
    v3 look_at = V(0.0f, 1.0f, 0.0f);
    v3 d = sub(look_at, V(pos[0], pos[1], pos[2]));
    v3 n = normalize(d);
    dir_out[0] = n.x; dir_out[1] = n.y; dir_out[2] = n.z;
    *focus_out = sqrtf(dot(d, d));
}

/* glm::normalize as setCamera applies it to the direction it stores,
 * src/rt/RayTracer.cpp:301 */
void rto_normalize3(const float in[3], float out[3])
{
    v3 n = normalize(V(in[0], in[1], in[2]));
    out[0] = n.x; out[1] = n.y; out[2] = n.z;
}

/* src/rt/RayTracer.cpp:140-177; dir is the STORED mCamDir (already normalised
 * once by create() or setCamera()); :143 normalises it again. */
void rto_make_camera_params(const float pos[3], const float dir[3],
                            float aperture, float vfov_deg, float focus,
                            uint32_t width, uint32_t height, rto_params *out)
{
    v3 look_from = V(pos[0], pos[1], pos[2]);
    v3 direction = normalize(V(dir[0], dir[1], dir[2])); /* :143 */
    v3 look_at = add(look_from, direction);
    v3 vup = V(0.0f, 1.0f, 0.0f);

    float aspect = (float)width / (float)height;
    float theta = vfov_deg * 0.01745329251994329576923690768489f; /* glm::radians */
    float half_height = tanf(theta * 0.5f);
    float viewport_height = 2.0f * half_height;
    float viewport_width = aspect * viewport_height;

    v3 w = normalize(sub(look_from, look_at));
    v3 u = normalize(cross(vup, w));
    v3 v = cross(w, u);

    v3 horizontal = scale(u, focus * viewport_width);
    v3 vertical = scale(v, focus * viewport_height);
    v3 lower_left = sub(sub(sub(look_from, scale(horizontal, 0.5f)), scale(vertical, 0.5f)), scale(w, focus));

    memset(out, 0, sizeof(*out));
    out->origin_lens[0] = look_from.x; out->origin_lens[1] = look_from.y; out->origin_lens[2] = look_from.z;
    out->origin_lens[3] = aperture * 0.5f;
    out->lower_left[0] = lower_left.x; out->lower_left[1] = lower_left.y; out->lower_left[2] = lower_left.z;
    out->horizontal[0] = horizontal.x; out->horizontal[1] = horizontal.y; out->horizontal[2] = horizontal.z;
    out->vertical[0] = vertical.x; out->vertical[1] = vertical.y; out->vertical[2] = vertical.z;
    out->u[0] = u.x; out->u[1] = u.y; out->u[2] = u.z;
    out->v[0] = v.x; out->v[1] = v.y; out->v[2] = v.z;
    out->w[0] = w.x; out->w[1] = w.y; out->w[2] = w.z;
    out->resolution[0] = (float)width; out->resolution[1] = (float)height;
    out->inv_resolution[0] = 1.0f / (float)width; out->inv_resolution[1] = 1.0f / (float)height;
}

/* ------------------------------------------------------------------ */
/* device half: raytrace.comp.glsl                                     */
/* ------------------------------------------------------------------ */
typedef struct { v3 o, d; } ray_t;

typedef struct {
    int hit;
    float t;
    v3 p, n;
    int front_face;
    uint32_t mat;
    float fuzz, ref_idx;
    v3 albedo;
    float flags;
} hit_t;

typedef struct {
    uint32_t rng;
    rto_counters *c; /* never NULL inside the tracer (a dummy is used) */
} ctx_t;

static inline float rnd(ctx_t *cx)
{
    cx->c->rand_draws++;
    return rto_rand(&cx->rng);
}

/* raytrace.comp.glsl:57-68 */
static v3 random_in_unit_sphere(ctx_t *cx)
{
    v3 p;
    do {
        float a = rnd(cx);
        float b = rnd(cx);
        float c = rnd(cx);
        p = V(a * 2.0f - 1.0f, b * 2.0f - 1.0f, c * 2.0f - 1.0f);
        cx->c->sphere_iters++;
    } while (dot(p, p) >= 1.0f);
    return p;
}

/* raytrace.comp.glsl:70-73 */
static v3 random_unit_vector(ctx_t *cx)
{
    return normalize(random_in_unit_sphere(cx));
}

/* raytrace.comp.glsl:75-86 */
static v3 random_in_unit_disk(ctx_t *cx)
{
    v3 p;
    do {
        float a = rnd(cx);
        float b = rnd(cx);
        p = V(a * 2.0f - 1.0f, b * 2.0f - 1.0f, 0.0f);
        cx->c->disk_iters++;
    } while ((p.x * p.x + p.y * p.y) >= 1.0f);
    return p;
}

/* raytrace.comp.glsl:109-153 */
static int hit_sphere(const rto_sphere *s, ray_t ray, float tmin, float tmax, hit_t *h, rto_counters *c)
{
    v3 center = V(s->center_radius[0], s->center_radius[1], s->center_radius[2]);
    float radius = s->center_radius[3];
    v3 oc = sub(ray.o, center);
    float a = dot(ray.d, ray.d);
    float half_b = dot(oc, ray.d);
    float cc = dot(oc, oc) - radius * radius;
    float discriminant = half_b * half_b - a * cc;

    c->sphere_tests++;
    if (discriminant < 0.0f) {
        return 0;
    }
    c->disc_nonneg++;

    float sqrt_disc = sqrtf(discriminant);
    float root = (-half_b - sqrt_disc) / a;

    if (root < tmin || root > tmax) {
        c->far_roots++;
        root = (-half_b + sqrt_disc) / a;
        if (root < tmin || root > tmax) {
            return 0;
        }
    }
    c->accepts++;

    h->t = root;
    h->p = add(ray.o, scale(ray.d, root));
    h->n = divs(sub(h->p, center), radius);
    h->front_face = dot(ray.d, h->n) < 0.0f;
    if (!h->front_face) {
        h->n = neg(h->n);
    }
    h->mat = (uint32_t)(s->misc[0] + 0.5f);
    h->fuzz = s->misc[1];
    h->ref_idx = s->misc[2];
    h->albedo = V(s->albedo[0], s->albedo[1], s->albedo[2]);
    h->flags = s->misc[3];
    h->hit = 1;
    return 1;
}

/* raytrace.comp.glsl:155-174 */
static int scatter_lambert(const hit_t *h, v3 *attenuation, ray_t *next, ctx_t *cx)
{
    v3 target = add(add(h->p, h->n), random_unit_vector(cx));
    next->o = add(h->p, scale(h->n, 0.001f));
    next->d = normalize(sub(target, h->p));

    v3 albedo = h->albedo;
    cx->c->lambert++;
    if (h->flags > 0.5f) {
        float checker_index = floorf(h->p.x) + floorf(h->p.z);
        float checker = glsl_mod(checker_index, 2.0f); /* :166 */
        float tone = (checker < 1.0f) ? 0.05f : 0.95f;
        albedo = scale(albedo, tone);
        cx->c->lambert_checker++;
    }
    *attenuation = mul(*attenuation, albedo);
    return 1;
}

/* raytrace.comp.glsl:176-190 */
static int refract_vec(v3 incident, v3 normal, float eta, v3 *refracted)
{
    float cos_theta = glsl_min(dot(neg(incident), normal), 1.0f);
    v3 refract_perp = scale(add(incident, scale(normal, cos_theta)), eta);
    float perp_dot = dot(refract_perp, refract_perp);
    if (perp_dot > 1.0f) {
        return 0;
    }
    v3 refract_parallel = scale(normal, -sqrtf(1.0f - perp_dot));
    *refracted = add(refract_perp, refract_parallel);
    return 1;
}

/* raytrace.comp.glsl:192-197 */
static float schlick(float cosine, float ref_idx)
{
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 *= r0;
    return r0 + (1.0f - r0) * rto_powf(1.0f - cosine, 5.0f);
}

/* raytrace.comp.glsl:199-208 */
static int scatter_metal(const hit_t *h, v3 *attenuation, ray_t *next, ctx_t *cx)
{
    v3 reflected = reflect(normalize(next->d), h->n);
    next->o = add(h->p, scale(h->n, 0.001f));
    next->d = add(reflected, scale(random_in_unit_sphere(cx), h->fuzz));
    *attenuation = mul(*attenuation, h->albedo);
    cx->c->metal++;
    int ok = dot(next->d, h->n) > 0.0f;
    if (!ok) {
        cx->c->metal_absorbed++;
    }
    return ok;
}

/* raytrace.comp.glsl:210-234 */
static int scatter_dielectric(const hit_t *h, v3 *attenuation, ray_t *next, ctx_t *cx)
{
    *attenuation = mul(*attenuation, V(1.0f, 1.0f, 1.0f));
    cx->c->dielectric++;

    float eta_ratio = h->front_face ? (1.0f / h->ref_idx) : h->ref_idx;
    if (h->front_face) {
        cx->c->diel_front++;
    }
    v3 unit_dir = normalize(next->d);
    float cos_theta = glsl_min(dot(neg(unit_dir), h->n), 1.0f);
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    v3 direction;
    int cannot_refract = eta_ratio * sin_theta > 1.0f;
    int do_reflect = cannot_refract;
    if (!do_reflect) {
        /* `||` short-circuits: the draw happens only when refraction is possible */
        cx->c->diel_schlick++;
        float sch = schlick(cos_theta, eta_ratio);
        do_reflect = sch > rnd(cx);
    }
    if (do_reflect) {
        direction = reflect(unit_dir, h->n);
    } else {
        cx->c->diel_refract++;
        if (!refract_vec(unit_dir, h->n, eta_ratio, &direction)) {
            direction = reflect(unit_dir, h->n);
        }
    }
    next->o = add(h->p, scale(direction, 0.001f));
    next->d = direction;
    return 1;
}

/* raytrace.comp.glsl:236-296 */
static v3 trace_ray(const rto_params *params, const rto_sphere *spheres, ray_t ray, ctx_t *cx)
{
    v3 throughput = V(1.0f, 1.0f, 1.0f);
    v3 radiance = V(0.0f, 0.0f, 0.0f);
    uint32_t max_depth = params->frame_sample_depth_count[2];
    uint32_t depth;

    for (depth = 0u; depth < max_depth; ++depth) {
        float closest = 1e20f;
        hit_t best;
        uint32_t sphere_count = params->frame_sample_depth_count[3];
        uint32_t i;
        memset(&best, 0, sizeof(best));
        cx->c->segments++;

        for (i = 0u; i < sphere_count; ++i) {
            hit_t cand;
            if (hit_sphere(&spheres[i], ray, 0.001f, closest, &cand, cx->c)) {
                closest = cand.t;
                best = cand;
            }
        }

        if (!best.hit) {
            v3 unit_dir = normalize(ray.d);
            float t = 0.5f * (unit_dir.y + 1.0f);
            v3 sky = mix3(V(1.0f, 1.0f, 1.0f), V(0.5f, 0.7f, 1.0f), t); /* :265 */
            radiance = add(radiance, mul(throughput, sky));
            cx->c->sky++;
            return radiance;
        }

        ray_t next = ray;
        int scattered;
        if (best.mat == 0u) {
            scattered = scatter_lambert(&best, &throughput, &next, cx);
        } else if (best.mat == 1u) {
            scattered = scatter_metal(&best, &throughput, &next, cx);
        } else {
            scattered = scatter_dielectric(&best, &throughput, &next, cx);
        }
        if (!scattered) {
            return radiance;
        }
        ray = next;
    }
    cx->c->depth_exhausted++;
    return radiance;
}

/* body of the spp loop, raytrace.comp.glsl:317-330 */
static v3 one_sample(const rto_params *pr, const rto_sphere *spheres, uint32_t px, uint32_t py, ctx_t *cx)
{
    float u_coord = ((float)(int32_t)px + rnd(cx)) * pr->inv_resolution[0];
    /* float(resolution.y - 1u - pixel.y): float arithmetic, left to right */
    float v_coord = (((pr->resolution[1] - 1.0f) - (float)(int32_t)py) + rnd(cx)) * pr->inv_resolution[1];

    v3 disk = random_in_unit_disk(cx);
    v3 random_disk = scale(disk, pr->origin_lens[3]);
    v3 cu = V(pr->u[0], pr->u[1], pr->u[2]);
    v3 cv = V(pr->v[0], pr->v[1], pr->v[2]);
    v3 offset = add(scale(cu, random_disk.x), scale(cv, random_disk.y));

    v3 origin = V(pr->origin_lens[0], pr->origin_lens[1], pr->origin_lens[2]);
    v3 ll = V(pr->lower_left[0], pr->lower_left[1], pr->lower_left[2]);
    v3 hz = V(pr->horizontal[0], pr->horizontal[1], pr->horizontal[2]);
    v3 vt = V(pr->vertical[0], pr->vertical[1], pr->vertical[2]);
    v3 dir = sub(sub(add(add(ll, scale(hz, u_coord)), scale(vt, v_coord)), origin), offset);

    ray_t ray;
    ray.o = add(origin, offset);
    ray.d = normalize(dir);
    cx->c->samples++;
    return trace_ray(pr, spheres, ray, cx);
}

static inline uint8_t unorm8(float c)
{
    float x = glsl_min(glsl_max(c, 0.0f), 1.0f);
    return (uint8_t)floorf(x * 255.0f + 0.5f);
}

/* The display tail, raytrace.comp.glsl:337-342, for one texel: total / sample_count, c / (c + 1),
 * pow(max(c, 0), 1/2.2), UNORM8 store. */
static void display_tail(v3 total, float sample_count, uint8_t *out4)
{
    v3 color = divs(total, sample_count);
    color = V(color.x / (color.x + 1.0f), color.y / (color.y + 1.0f), color.z / (color.z + 1.0f));
    /* vec3(1.0 / 2.2): the GLSL front end folds the constant in double */
    const float inv_gamma = (float)(1.0 / 2.2);
    out4[0] = unorm8(rto_powf(glsl_max(color.x, 0.0f), inv_gamma));
    out4[1] = unorm8(rto_powf(glsl_max(color.y, 0.0f), inv_gamma));
    out4[2] = unorm8(rto_powf(glsl_max(color.z, 0.0f), inv_gamma));
    out4[3] = 255;
}

void rto_resolve(const float *accum, uint8_t *out_rgba8, size_t texels, uint32_t sample_count)
{
    size_t i;
    for (i = 0; i < texels; ++i) {
        display_tail(V(accum[4 * i], accum[4 * i + 1], accum[4 * i + 2]), (float)sample_count, out_rgba8 + 4 * i);
    }
}

/* raytrace.comp.glsl:298-343 for one pixel */
static void shade_pixel(const rto_params *pr, const rto_sphere *spheres, uint32_t px, uint32_t py,
                        uint32_t width, float *accum, uint8_t *out8, uint32_t *rng_final, rto_counters *c)
{
    uint32_t frame = pr->frame_sample_depth_count[0];
    uint32_t spp_raw = pr->frame_sample_depth_count[1];
    uint32_t spp = spp_raw > 1u ? spp_raw : 1u;
    ctx_t cx;
    v3 acc = V(0.0f, 0.0f, 0.0f);
    uint32_t s;
    size_t idx = ((size_t)py * width + px);

    cx.rng = rto_hash3(px, py, frame);
    cx.c = c;
    c->pixels++;

    for (s = 0u; s < spp; ++s) {
        acc = add(acc, one_sample(pr, spheres, px, py, &cx));
    }

    float *texel = accum + idx * 4;
    v3 total = add(V(texel[0], texel[1], texel[2]), acc);
    texel[0] = total.x; texel[1] = total.y; texel[2] = total.z; texel[3] = 1.0f;

    if (rng_final) {
        rng_final[idx] = cx.rng;
    }
    if (out8) {
        /* (frame + 1u) * spp wraps in uint32 before the conversion */
        display_tail(total, (float)((frame + 1u) * spp_raw), out8 + idx * 4);
    }
}

typedef struct {
    const rto_params *pr;
    const rto_sphere *spheres;
    float *accum;
    uint8_t *out8;
    uint32_t *rng_final;
    uint32_t y0, y1, width;
    volatile uint32_t *next_row;
    rto_counters counters;
} job_t;

static void *worker(void *arg)
{
    job_t *j = (job_t *)arg;
    for (;;) {
        uint32_t y = __sync_fetch_and_add(j->next_row, 1u);
        uint32_t x;
        if (y >= j->y1) {
            break;
        }
        for (x = 0; x < j->width; ++x) {
            shade_pixel(j->pr, j->spheres, x, y, j->width, j->accum, j->out8, j->rng_final, &j->counters);
        }
    }
    return NULL;
}

static void add_counters(rto_counters *dst, const rto_counters *src)
{
    const uint64_t *s = (const uint64_t *)src;
    uint64_t *d = (uint64_t *)dst;
    size_t i;
    for (i = 0; i < sizeof(rto_counters) / sizeof(uint64_t); ++i) {
        d[i] += s[i];
    }
}

int rto_render(const rto_params *params, const rto_sphere *spheres,
               float *accum, uint8_t *out_rgba8, uint32_t *rng_final,
               uint32_t y0, uint32_t y1, int nthreads, rto_counters *counters)
{
    /* bounds: the shader compares against int(resolution) */
    uint32_t width = (uint32_t)(int32_t)params->resolution[0];
    uint32_t height = (uint32_t)(int32_t)params->resolution[1];
    volatile uint32_t next_row = y0;
    int t;
    if (y1 > height) {
        y1 = height;
    }
    if (nthreads <= 0) {
        nthreads = 1;
    }
    if (nthreads > 256) {
        nthreads = 256;
    }
    job_t *jobs = (job_t *)calloc((size_t)nthreads, sizeof(job_t));
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    for (t = 0; t < nthreads; ++t) {
        jobs[t].pr = params; jobs[t].spheres = spheres; jobs[t].accum = accum;
        jobs[t].out8 = out_rgba8; jobs[t].rng_final = rng_final;
        jobs[t].y0 = y0; jobs[t].y1 = y1; jobs[t].width = width; jobs[t].next_row = &next_row;
    }
    if (nthreads == 1) {
        worker(&jobs[0]);
    } else {
        for (t = 0; t < nthreads; ++t) {
            pthread_create(&th[t], NULL, worker, &jobs[t]);
        }
        for (t = 0; t < nthreads; ++t) {
            pthread_join(th[t], NULL);
        }
    }
    if (counters) {
        for (t = 0; t < nthreads; ++t) {
            add_counters(counters, &jobs[t].counters);
        }
    }
    free(jobs);
    free(th);
    return 0;
}

/* Test hook: the helpers above by name, so that fixtures produced with the reference's vendored GLM
 * can pin them.  Returns the number of floats written to out, or -1 for an unknown name. */
int rto_builtin(const char *op, const float *in, float *out)
{
    v3 a = V(in[0], in[1], in[2]);
    if (strcmp(op, "dot") == 0) { out[0] = dot(a, V(in[3], in[4], in[5])); return 1; }
    if (strcmp(op, "normalize") == 0) { v3 r = normalize(a); out[0] = r.x; out[1] = r.y; out[2] = r.z; return 3; }
    if (strcmp(op, "reflect") == 0) { v3 r = reflect(a, V(in[3], in[4], in[5])); out[0] = r.x; out[1] = r.y; out[2] = r.z; return 3; }
    if (strcmp(op, "cross") == 0) { v3 r = cross(a, V(in[3], in[4], in[5])); out[0] = r.x; out[1] = r.y; out[2] = r.z; return 3; }
    if (strcmp(op, "length") == 0) { out[0] = sqrtf(dot(a, a)); return 1; }
    if (strcmp(op, "mix") == 0) { v3 r = mix3(a, V(in[3], in[4], in[5]), in[6]); out[0] = r.x; out[1] = r.y; out[2] = r.z; return 3; }
    if (strcmp(op, "mod") == 0) { out[0] = glsl_mod(in[0], in[1]); return 1; }
    if (strcmp(op, "floor") == 0) { out[0] = floorf(in[0]); return 1; }
    if (strcmp(op, "inversesqrt") == 0) { out[0] = inversesqrt(in[0]); return 1; }
    if (strcmp(op, "min") == 0) { out[0] = glsl_min(in[0], in[1]); return 1; }
    if (strcmp(op, "max") == 0) { out[0] = glsl_max(in[0], in[1]); return 1; }
    if (strcmp(op, "pow") == 0) { out[0] = rto_powf(in[0], in[1]); return 1; }
    return -1;
}

int rto_hit_sphere(const rto_sphere *s, const float o[3], const float d[3],
                   float tmin, float tmax, float *t, float p[3], float n[3], int *front)
{
    rto_counters dummy;
    hit_t h;
    ray_t r;
    memset(&dummy, 0, sizeof(dummy));
    memset(&h, 0, sizeof(h));
    r.o = V(o[0], o[1], o[2]);
    r.d = V(d[0], d[1], d[2]);
    if (!hit_sphere(s, r, tmin, tmax, &h, &dummy)) {
        return 0;
    }
    *t = h.t;
    p[0] = h.p.x; p[1] = h.p.y; p[2] = h.p.z;
    n[0] = h.n.x; n[1] = h.n.y; n[2] = h.n.z;
    *front = h.front_face;
    return 1;
}

void rto_sample(const rto_params *params, const rto_sphere *spheres,
                uint32_t px, uint32_t py, uint32_t *rng, float rgb[3])
{
    rto_counters dummy;
    ctx_t cx;
    v3 r;
    memset(&dummy, 0, sizeof(dummy));
    cx.rng = *rng;
    cx.c = &dummy;
    r = one_sample(params, spheres, px, py, &cx);
    *rng = cx.rng;
    rgb[0] = r.x; rgb[1] = r.y; rgb[2] = r.z;
}

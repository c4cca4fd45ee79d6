// Device-side arithmetic of the sphere path tracer (gfx950).
//
// Every function here is the HIP counterpart of one function of the reference
// compute shader shaders/raytrace.comp.glsl; the line ranges are given at each
// definition.  The translation unit that includes this header MUST be compiled
// with -ffp-contract=off and without fast-math: the per-pixel random streams
// are steered by float comparisons, so every +,-,*,/ and sqrt below has to be
// one correctly rounded IEEE-754 binary32 operation, in shader source order.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt {

struct f3 {
    float x, y, z;
};

__device__ __forceinline__ f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 operator/(f3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ f3 operator-(f3 a) { return mk(-a.x, -a.y, -a.z); }
// GLSL dot(): products summed left to right.
__device__ __forceinline__ float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// GLSL min/max: "y if y < x, otherwise x" / "y if x < y, otherwise x".
__device__ __forceinline__ float glsl_min(float x, float y) { return (y < x) ? y : x; }
__device__ __forceinline__ float glsl_max(float x, float y) { return (x < y) ? y : x; }
// ---------------------------------------------------------------- exact divide / sqrt, cheaper
// `x / y` and sqrtf() compile to the full IEEE sequences (v_div_scale, v_rcp, five fma, v_div_fmas,
// v_div_fixup; v_sqrt plus input scaling, a one-ulp correction and a class fix-up).  The scaling and
// fix-up steps only matter for operands near the ends of the exponent range, and the denominator
// half of a division (reciprocal + one Newton step) can be shared by every division with the same
// denominator.  The helpers below run the same core arithmetic without those steps when the
// operands are in a range where they are no-ops, and fall back to the full sequence otherwise, so
// the result is the correctly rounded one in every case.
// The fall-back blocks start with an empty `asm volatile`: it cannot be executed speculatively, so
// the compiler keeps them behind a branch that is skipped when no lane needs them, instead of
// computing both variants and selecting.
#define RT_COLD_PATH() asm volatile("; full-range sequence" ::)
// The same device for short blocks that only assign: behind a real branch the assignments are
// v_mov under the branch's exec mask.  If-converted they become v_cndmask_b32 reading a VCC that a
// scalar instruction combined, and on gfx950 such a select costs about ten times a v_mov (the VOP2
// select is only cheap right behind the vector compare that wrote VCC).
#define RT_KEEP_BRANCH() asm volatile("; assignments under exec" ::)
constexpr float kDenLo = 0x1p-20f;  // denominators in [2^-20, 2^20]
constexpr float kDenHi = 0x1p20f;
constexpr float kNumLo = 0x1p-90f;  // numerators with 2^-90 <= |n| < 2^100
constexpr float kNumHi = 0x1p100f;

// lo <= x <= hi for positive finite lo <= hi, as one unsigned comparison of bit patterns (positive floats
// order like their bits; a negative x, a NaN or an infinity above hi falls outside like in the two
// float comparisons).
__device__ __forceinline__ bool in_closed_range(float x, float lo, float hi)
{
    return __float_as_uint(x) - __float_as_uint(lo) <= __float_as_uint(hi) - __float_as_uint(lo);
}
// lo <= x < hi
__device__ __forceinline__ bool in_half_open_range(float x, float lo, float hi)
{
    return __float_as_uint(x) - __float_as_uint(lo) < __float_as_uint(hi) - __float_as_uint(lo);
}
__device__ __forceinline__ bool den_in_range(float d) { return in_closed_range(d, kDenLo, kDenHi); }
__device__ __forceinline__ bool num_in_range(float n) { return fabsf(n) >= kNumLo && fabsf(n) < kNumHi; }

// Denominator half of the division sequence: rcp and one Newton step.
__device__ __forceinline__ float refined_rcp(float d)
{
    float r0 = __builtin_amdgcn_rcpf(d);
    float e = __builtin_fmaf(-d, r0, 1.0f);
    return __builtin_fmaf(e, r0, r0);
}
// Numerator half: quotient estimate, two residual corrections.
__device__ __forceinline__ float div_core(float n, float d, float r)
{
    float q = n * r;
    float e = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(e, r, q);
}
// n / d given r = refined_rcp(d) and d_ok = den_in_range(d).
__device__ __forceinline__ float div_shared(float n, float d, float r, bool d_ok)
{
    float q = div_core(n, d, r);
    if (!(d_ok && num_in_range(n))) {
        RT_COLD_PATH();
        q = n / d;
    }
    return q;
}
// Correctly rounded sqrt for 2^-90 <= x < 2^100 (v_sqrt_f32 is within one ulp; pick the neighbour
// whose residual changes sign), full sequence otherwise.
__device__ __forceinline__ float sqrt_core(float x)
{
    // s is within one ulp of sqrt(x).  With s_dn / s_up its neighbours, the correctly rounded root is
    // s_dn if x - s_dn*s <= 0, s_up if x - s_up*s > 0, else s (the two cannot hold together).  The
    // residuals are formed with the opposite sign, e_dn = s_dn*s - x and e_up = s_up*s - x (exact
    // zeros come out as +0), so each condition is one sign bit: e_dn >= 0 and e_up < 0.
    const float s = __builtin_amdgcn_sqrtf(x);
    const uint32_t s_dn_bits = __float_as_uint(s) - 1u;
    const float s_dn = __uint_as_float(s_dn_bits);
    const float s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float e_dn = __builtin_fmaf(s_dn, s, -x);
    const float e_up = __builtin_fmaf(s_up, s, -x);
    const int32_t keep = ((int32_t)__float_as_uint(e_dn)) >> 31; // -1: e_dn < 0, stay at s;  0: step down
    const int32_t up = ((int32_t)__float_as_uint(e_up)) >> 31;   // -1: e_up < 0, step up;    0: stay
    // s_dn_bits + (keep ? 1 : 0) + (up ? 1 : 0)
    return __uint_as_float(s_dn_bits - (uint32_t)keep - (uint32_t)up);
}
__device__ __forceinline__ float sqrt_exact(float x)
{
    float s = sqrt_core(x);
    if (!in_half_open_range(x, kNumLo, kNumHi)) {
        RT_COLD_PATH();
        s = sqrtf(x);
    }
    return s;
}
// inversesqrt as two correctly rounded steps (sqrt, then 1/s); normalize(v) = v * inversesqrt(dot(v,v)).
__device__ __forceinline__ float inversesqrt(float x)
{
    // 2^-40 <= x <= 2^40 keeps both the sqrt and the reciprocal of its result inside the cheap ranges
    float s = sqrt_core(x);
    // 1 / s: on gfx950 v_rcp_f32 and one Newton step give the correctly rounded reciprocal of every s
    // that can occur here (tools/exhaustive_inversesqrt.py compares this function with the full
    // sequences for every binary32 x in [2^-40, 2^40]: no difference in 671 088 641 values; a residual
    // step on top changes nothing)
    float q = refined_rcp(s);
    if (!in_closed_range(x, 0x1p-40f, 0x1p40f)) {
        RT_COLD_PATH();
        q = 1.0f / sqrtf(x);
    }
    return q;
}
__device__ __forceinline__ f3 normalize_with_dot(f3 a, float d2) { return a * inversesqrt(d2); }
__device__ __forceinline__ f3 normalize(f3 a) { return a * inversesqrt(dot(a, a)); }
// Hand-placed packed FP32 (experiment, -DRT_PK_SHADE): the x and y components of a vec3 operation as one
// v_pk_mul_f32 / v_pk_add_f32 on a register pair, z as a scalar operation.  Same roundings, no contraction.
typedef float pk2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f3 pk_mul(f3 a, f3 b)
{
    const pk2 r = pk2{ a.x, a.y } * pk2{ b.x, b.y };
    return mk(r.x, r.y, a.z * b.z);
}
__device__ __forceinline__ f3 pk_scale_add(f3 a, f3 b, float s) // a + b * s
{
    const pk2 t = pk2{ b.x, b.y } * pk2{ s, s };
    const pk2 r = pk2{ a.x, a.y } + t;
    return mk(r.x, r.y, a.z + b.z * s);
}
__device__ __forceinline__ f3 pk_scale_sub(f3 a, f3 b, float s) // a - b * s
{
    const pk2 t = pk2{ b.x, b.y } * pk2{ s, s };
    const pk2 r = pk2{ a.x, a.y } - t;
    return mk(r.x, r.y, a.z - b.z * s);
}
// reflect(I, N) = I - 2.0 * dot(N, I) * N
#ifdef RT_PK_SHADE
__device__ __forceinline__ f3 reflect(f3 i, f3 n) { return pk_scale_sub(i, n, 2.0f * dot(n, i)); }
#else
__device__ __forceinline__ f3 reflect(f3 i, f3 n) { return i - n * (2.0f * dot(n, i)); }
#endif

// ---------------------------------------------------------------- RNG
// raytrace.comp.glsl:37-44 (one component of the uvec3 hash)
__device__ __forceinline__ uint32_t hash1(uint32_t v)
{
    v = (v ^ (v >> 17)) * 0xED5AD4BBu;
    v = (v ^ (v >> 11)) * 0xAC4C1B51u;
    v = (v ^ (v >> 15)) * 0x31848BABu;
    return v;
}
__device__ __forceinline__ uint32_t hash3(uint32_t x, uint32_t y, uint32_t z)
{
    return hash1(x) ^ hash1(y) ^ hash1(z);
}
// One step of the generator of raytrace.comp.glsl:46-50, a * s + c mod 2^32.  With a compile-time
// increment the step is one v_mad_u64_u32 (32 x 32 + 64 -> 64 bits, of which the low word is a * s + c
// whatever the high word of the addend holds; the increment sits in a scalar register pair) instead of
// v_mul_lo_u32 + v_add_u32: the multiply-add issues as fast as the multiply alone, +2.4 % on the benchmark.
// (With a per-lane increment the addend needs a vector register pair, which costs the kernel its eighth
// wave per SIMD; that form is left to the compiler.)
__device__ __forceinline__ uint32_t lcg_step(uint32_t s, uint32_t a, uint32_t c)
{
    if (__builtin_constant_p(c)) {
        unsigned long long wide;
        const unsigned long long addend = c;
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(wide) : "v"(s), "v"(a), "s"(addend) : "vcc");
        return (uint32_t)wide;
    }
    return a * s + c;
}
// raytrace.comp.glsl:46-50
__device__ __forceinline__ uint32_t lcg(uint32_t& state)
{
    state = lcg_step(state, 1664525u, 1013904223u);
    return state;
}
// raytrace.comp.glsl:52-55.  The divide by 2^24 is exact, so a multiply by
// 2^-24 produces the same bits.
__device__ __forceinline__ float rnd(uint32_t& state)
{
    return (float)(lcg(state) & 0x00FFFFFFu) * 5.9604644775390625e-08f;
}
// base + rand(state) in one instruction.  Multiplying the 24-bit integer by 2^-24 is exact, so the
// fused multiply-add rounds the same real number as the separate addition does.  (The same holds
// wherever this file or the kernel fuses a product with a power of two; products that round are
// never fused.)
__device__ __forceinline__ float rnd_plus(uint32_t& state, float base)
{
    return __builtin_fmaf((float)(lcg(state) & 0x00FFFFFFu), 5.9604644775390625e-08f, base);
}

// ---------------------------------------------------------------- pow
// GLSL pow(x,y) inherits its precision from exp2(y*log2(x)).  These kernels use
// only +,-,*,/ and integer bit operations in a fixed order, so the host-side
// checker reproduces them bit for bit (schlick :196 feeds a comparison that
// steers the random stream).
__device__ __forceinline__ float portable_log2(float x)
{
    uint32_t ix = __float_as_uint(x);
    int k = 0;
    if (ix >= 0x7f800000u) {
        return (ix > 0x80000000u && ix <= 0xff800000u) ? __uint_as_float(0x7fc00000u) : x;
    }
    if (ix == 0u) {
        return __uint_as_float(0xff800000u);
    }
    if (ix < 0x00800000u) {
        x = x * 8388608.0f;
        k = -23;
        ix = __float_as_uint(x);
    }
    k += (int)(ix >> 23) - 127;
    ix &= 0x007fffffu;
    uint32_t i = (ix + 0x004afb20u) & 0x00800000u;
    float m = __uint_as_float(ix | (i ^ 0x3f800000u));
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

__device__ __forceinline__ float pow2i(int n) { return __uint_as_float((uint32_t)(n + 127) << 23); }

__device__ __forceinline__ float portable_exp2(float p)
{
    if (!(p == p)) {
        return p;
    }
    if (p >= 128.0f) {
        return __uint_as_float(0x7f800000u);
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

__device__ __forceinline__ float glsl_pow(float x, float y)
{
    if (x == 0.0f) {
        return (y > 0.0f) ? 0.0f : ((y == 0.0f) ? 1.0f : __uint_as_float(0x7f800000u));
    }
    return portable_exp2(y * portable_log2(x));
}

// ---------------------------------------------------------------- samplers
// raytrace.comp.glsl:57-68
__device__ __forceinline__ f3 random_in_unit_sphere(uint32_t& rng)
{
    f3 p;
    do {
        float a = rnd(rng);
        float b = rnd(rng);
        float c = rnd(rng);
        p = mk(a * 2.0f - 1.0f, b * 2.0f - 1.0f, c * 2.0f - 1.0f);
    } while (dot(p, p) >= 1.0f);
    return p;
}

// raytrace.comp.glsl:75-86 (z is always 0)
__device__ __forceinline__ void random_in_unit_disk(uint32_t& rng, float& dx, float& dy)
{
    do {
        float a = rnd(rng);
        float b = rnd(rng);
        dx = a * 2.0f - 1.0f;
        dy = b * 2.0f - 1.0f;
    } while ((dx * dx + dy * dy) >= 1.0f);
}

// ---------------------------------------------------------------- intersection
// Per-ray constants of the intersection: a = dot(d,d) (the same for every sphere, :112), the shared
// denominator half of the two root divisions, and whether `a` is in the cheap-division range.
struct RayDenominator {
    float a;
    float rcp_a;
    bool a_ok;
    // a_ok, the origin below 2^30 in magnitude and every sphere within the bounded range
    // (|centre| <= 2^20, radius^2 <= 2^40, all finite): half_b, c and the discriminant of every
    // sphere are then finite and far from overflow, which the sign test of behind_origin() needs
    bool bounded;
};
__device__ __forceinline__ RayDenominator ray_denominator(f3 d)
{
    RayDenominator r;
    r.a = dot(d, d);
    r.rcp_a = refined_rcp(r.a);
    r.a_ok = den_in_range(r.a);
    r.bounded = false;
    return r;
}
__device__ __forceinline__ RayDenominator ray_denominator(f3 o, f3 d, bool scene_bounded)
{
    RayDenominator r = ray_denominator(d);
    r.bounded = scene_bounded && r.a_ok && (fabsf(o.x) + fabsf(o.y)) + fabsf(o.z) < 0x1p30f;
    return r;
}
// Whether the centre of a sphere (or of a ball) is behind the origin of a ray that starts outside
// it: half_b >= 0 and c >= 0, i.e. neither sign bit is set.  With a non-negative discriminant this
// gives sqrt(disc) <= half_b (disc <= fl(half_b^2) because a*c >= 0, and the correctly rounded
// square root of fl(x*x) is x), so both roots are <= 0, apart from a positive far root below 2^-43
// when half_b^2 underflows; either way they are below tmin > 0 and :125-131 reject them.
__device__ __forceinline__ bool behind_origin(float half_b, float c)
{
    return (int32_t)(__float_as_uint(half_b) | __float_as_uint(c)) >= 0;
}
__device__ __forceinline__ bool sphere_bounded(float4 centre_radius)
{
    return fabsf(centre_radius.x) <= 0x1p20f && fabsf(centre_radius.y) <= 0x1p20f && fabsf(centre_radius.z) <= 0x1p20f &&
           fabsf(centre_radius.w) <= 0x1p20f;
}
// (-half_b -/+ sqrt_disc) / a, :123 and :127.  A numerator below 2^-90 in magnitude gives a root
// far below tmin = 0.001 whichever way it is rounded, and such a root is rejected and never
// stored, so only the upper end of the numerator range needs the full division.
__device__ __forceinline__ float root_div(float n, const RayDenominator& rd)
{
    float q = div_core(n, rd.a, rd.rcp_a);
    if (!(rd.a_ok && fabsf(n) < kNumHi)) {
        RT_COLD_PATH();
        q = n / rd.a;
    }
    return q;
}

// hitSphere up to the discriminant, raytrace.comp.glsl:111-115.  geom = (centre, radius^2).
struct SphereQuadratic {
    float half_b;
    float c;
    float discriminant;
};
__device__ __forceinline__ SphereQuadratic sphere_quadratic(float4 geom, f3 o, f3 d, float a)
{
    SphereQuadratic q;
    f3 oc = o - mk(geom.x, geom.y, geom.z);
    q.half_b = dot(oc, d);
    q.c = dot(oc, oc) - geom.w;
    q.discriminant = q.half_b * q.half_b - a * q.c;
    return q;
}
// The same with dot(oc,oc) multiplied by `scale` first: scale = 1 reproduces sphere_quadratic bit
// for bit (x * 1.0f is x), a scale slightly below 1 is the conservative test of a bounding sphere.
__device__ __forceinline__ SphereQuadratic sphere_quadratic_scaled(float4 geom, f3 o, f3 d, float a, float scale)
{
    SphereQuadratic q;
    f3 oc = o - mk(geom.x, geom.y, geom.z);
    q.half_b = dot(oc, d);
    q.c = dot(oc, oc) * scale - geom.w;
    q.discriminant = q.half_b * q.half_b - a * q.c;
    return q;
}
// hitSphere from the discriminant test to the accepted root, :117-133.  Returns true and the
// accepted root when the sphere is hit inside [tmin, tmax]; tmin must be > 0.
template <class Stats>
__device__ __forceinline__ bool sphere_root_from(const SphereQuadratic& q, const RayDenominator& rden, float tmin, float tmax,
                                                 float& root_out, Stats& st)
{
    if (q.discriminant < 0.0f) {
        return false;
    }
    // origin outside the sphere and moving away from it: both roots are below tmin
    if (rden.bounded && behind_origin(q.half_b, q.c)) {
        return false;
    }
    st.section(5 /* ST_ROOT */);
    float sqrt_disc = sqrt_exact(q.discriminant);
    float root = root_div(-q.half_b - sqrt_disc, rden);
    if (root < tmin || root > tmax) {
        st.section(6 /* ST_FAR_ROOT */);
        root = root_div(-q.half_b + sqrt_disc, rden);
        if (root < tmin || root > tmax) {
            return false;
        }
    }
    root_out = root;
    return true;
}
// The same, folding the caller's bookkeeping of :254-258 (closest = t, best = this sphere) into the
// accepting paths, so that a sphere nobody hits costs no select instructions.
template <class Stats>
__device__ __forceinline__ void sphere_closest(const SphereQuadratic& q, const RayDenominator& rden, float tmin, int index,
                                               float& closest, int& best, Stats& st)
{
    if (q.discriminant < 0.0f) {
        return;
    }
    if (rden.bounded && behind_origin(q.half_b, q.c)) {
        return; // outside and moving away: both roots are below tmin
    }
    st.section(5 /* ST_ROOT */);
    const float sqrt_disc = sqrt_exact(q.discriminant);
    const float near_root = root_div(-q.half_b - sqrt_disc, rden);
    if (near_root < tmin || near_root > closest) {
        st.section(6 /* ST_FAR_ROOT */);
        const float far_root = root_div(-q.half_b + sqrt_disc, rden);
        if (!(far_root < tmin || far_root > closest)) {
            RT_KEEP_BRANCH();
            closest = far_root;
            best = index;
        }
    } else {
        closest = near_root;
        best = index;
    }
}
// The same for spheres that are not visited in index order.  The loop of :247-259 ends with the
// smallest accepted root and, among spheres whose roots are exactly equal, with the one tested
// last, i.e. the highest index (a root equal to `closest` is not rejected by :125); both are
// properties of the set of spheres, not of the order, so any order gives the loop's result once
// ties go to the higher index.  The caller has already applied the discriminant and
// behind-the-origin tests.
template <class Stats>
__device__ __forceinline__ void sphere_closest_any_order(const SphereQuadratic& q, const RayDenominator& rden, float tmin,
                                                         int index, float& closest, int& best, Stats& st)
{
    st.section(5 /* ST_ROOT */);
    const float sqrt_disc = sqrt_exact(q.discriminant);
    float root = root_div(-q.half_b - sqrt_disc, rden);
    if (root > closest) {
        // The far root is no smaller than the near one (the square root is >= 0 and a correctly rounded division by
        // a > 0 is monotonic), so :127-129 would reject it as well: a sphere behind the closest hit costs one root.
        return;
    }
    if (root < tmin) {
        st.section(6 /* ST_FAR_ROOT */);
        root = root_div(-q.half_b + sqrt_disc, rden);
        if (root < tmin || root > closest) {
            return;
        }
    }
    if (root < closest || index > best) {
        RT_KEEP_BRANCH();
        closest = root;
        best = index;
    }
}
// The root-finding half of hitSphere, raytrace.comp.glsl:111-133.
template <class Stats>
__device__ __forceinline__ bool sphere_root(float4 geom, f3 o, f3 d, const RayDenominator& rden, float tmin, float tmax,
                                            float& root_out, Stats& st)
{
    return sphere_root_from(sphere_quadratic(geom, o, d, rden.a), rden, tmin, tmax, root_out, st);
}

// The record-filling half of hitSphere, raytrace.comp.glsl:135-143.  rr = refined_rcp(radius), or NaN
// when the radius is outside the range of the short division sequence.
__device__ __forceinline__ void hit_finish_with(f3 centre, float radius, float rr, f3 o, f3 d, float t, f3& p, f3& n, bool& front)
{
    p = o + d * t;
    const f3 pc = p - centre;
    // n = (p - centre) / radius: three divisions by one denominator
    const bool r_ok = rr == rr;
    const float lo = fminf(fminf(fabsf(pc.x), fabsf(pc.y)), fabsf(pc.z));
    const float hi = fmaxf(fmaxf(fabsf(pc.x), fabsf(pc.y)), fabsf(pc.z));
    n = mk(div_core(pc.x, radius, rr), div_core(pc.y, radius, rr), div_core(pc.z, radius, rr));
    if (!(r_ok && lo >= kNumLo && hi < kNumHi)) {
        RT_COLD_PATH();
        n = pc / radius;
    }
    front = dot(d, n) < 0.0f;
    if (!front) {
        n = -n;
    }
}

__device__ __forceinline__ void hit_finish(f3 centre, float radius, f3 o, f3 d, float t, f3& p, f3& n, bool& front)
{
    hit_finish_with(centre, radius, den_in_range(radius) ? refined_rcp(radius) : __builtin_nanf(""), o, d, t, p, n, front);
}

// schlick, raytrace.comp.glsl:192-197
__device__ __forceinline__ float schlick(float cosine, float ref_idx)
{
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 *= r0;
    return r0 + (1.0f - r0) * glsl_pow(1.0f - cosine, 5.0f);
}

// refractVec, raytrace.comp.glsl:176-190
__device__ __forceinline__ bool refract_vec(f3 incident, f3 normal, float eta, f3& refracted)
{
    float cos_theta = glsl_min(dot(-incident, normal), 1.0f);
    f3 refract_perp = (incident + normal * cos_theta) * eta;
    float perp_dot = dot(refract_perp, refract_perp);
    if (perp_dot > 1.0f) {
        return false;
    }
    f3 refract_parallel = normal * (-sqrtf(1.0f - perp_dot));
    refracted = refract_perp + refract_parallel;
    return true;
}

// rgba8 UNORM store conversion of imageStore(outputImage, ...), raytrace.comp.glsl:342
__device__ __forceinline__ uint32_t unorm8(float c)
{
    float x = glsl_min(glsl_max(c, 0.0f), 1.0f);
    return (uint32_t)floorf(x * 255.0f + 0.5f);
}

// Display tail, raytrace.comp.glsl:337-342, for one texel.
__device__ __forceinline__ uint32_t tonemap_rgba8(f3 total, float sample_count)
{
    f3 color = total / sample_count;
    color = mk(color.x / (color.x + 1.0f), color.y / (color.y + 1.0f), color.z / (color.z + 1.0f));
    const float inv_gamma = (float)(1.0 / 2.2);
    uint32_t r = unorm8(glsl_pow(glsl_max(color.x, 0.0f), inv_gamma));
    uint32_t g = unorm8(glsl_pow(glsl_max(color.y, 0.0f), inv_gamma));
    uint32_t b = unorm8(glsl_pow(glsl_max(color.z, 0.0f), inv_gamma));
    return r | (g << 8) | (b << 16) | (255u << 24);
}

} // namespace rt

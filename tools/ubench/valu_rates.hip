// Micro-benchmark: issue cost of single VALU instructions on gfx950, in cycles per wave64
// instruction per SIMD, at a given number of waves per SIMD.  Diagnostic tool only.
//   hipcc --offload-arch=gfx950 -O2 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int kUnroll = 64;

#define BODY8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define BODY64(OP) BODY8(OP) BODY8(OP) BODY8(OP) BODY8(OP) BODY8(OP) BODY8(OP) BODY8(OP) BODY8(OP)

#define DEFINE_KERNEL(NAME, ASMSTR) DEFINE_KERNEL_C(NAME, ASMSTR, "vcc", "scc", "s20", "s21", "v10", "v11")
#define DEFINE_KERNEL_C(NAME, ASMSTR, ...)                                                     \
    __global__ __launch_bounds__(256) void k_##NAME(float* out, int iters, float seed)          \
    {                                                                                          \
        float r[8];                                                                            \
        for (int i = 0; i < 8; ++i) r[i] = seed + threadIdx.x * 1e-3f + i;                     \
        float a = seed * 0.5f + 1.0f, b = seed + 3.0f;                                         \
        for (int it = 0; it < iters; ++it) {                                                   \
            _Pragma("unroll") for (int u = 0; u < kUnroll / 8; ++u) {                          \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                \
                    asm volatile(ASMSTR : "+v"(r[i]) : "v"(a), "v"(b) : __VA_ARGS__);           \
                }                                                                              \
            }                                                                                  \
        }                                                                                      \
        float s = 0;                                                                           \
        for (int i = 0; i < 8; ++i) s += r[i];                                                 \
        if (s == 12345.678f) out[0] = s;                                                       \
    }

DEFINE_KERNEL(add_f32, "v_add_f32 %0, %0, %1")
DEFINE_KERNEL(mul_f32, "v_mul_f32 %0, %0, %1")
DEFINE_KERNEL(fma_f32, "v_fma_f32 %0, %0, %1, %2")
DEFINE_KERNEL(mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
DEFINE_KERNEL(mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
DEFINE_KERNEL(mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
DEFINE_KERNEL(mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
DEFINE_KERNEL(add_u32, "v_add_u32 %0, %0, %1")
DEFINE_KERNEL(and_b32, "v_and_b32 %0, %0, %1")
DEFINE_KERNEL(lshl_add_u32, "v_lshl_add_u32 %0, %0, 3, %1")
DEFINE_KERNEL(cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
DEFINE_KERNEL(cvt_u32_f32, "v_cvt_u32_f32 %0, %0")
DEFINE_KERNEL(rcp_f32, "v_rcp_f32 %0, %0")
DEFINE_KERNEL(sqrt_f32, "v_sqrt_f32 %0, %0")
DEFINE_KERNEL(rsq_f32, "v_rsq_f32 %0, %0")
DEFINE_KERNEL(div_scale, "v_div_scale_f32 %0, vcc, %0, %1, %0")
DEFINE_KERNEL(div_fmas, "v_div_fmas_f32 %0, %0, %1, %2")
DEFINE_KERNEL(div_fixup, "v_div_fixup_f32 %0, %0, %1, %2")
DEFINE_KERNEL(cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEFINE_KERNEL(cmp_lt, "v_cmp_lt_f32 vcc, %0, %1")
DEFINE_KERNEL(floor_f32, "v_floor_f32 %0, %0")
DEFINE_KERNEL(ldexp_f32, "v_ldexp_f32 %0, %0, %1")
DEFINE_KERNEL(bfe_i32, "v_bfe_i32 %0, %0, 0, 24")
DEFINE_KERNEL(max_f32, "v_max_f32 %0, %0, %1")
DEFINE_KERNEL(mov, "v_mov_b32 %0, %1")
DEFINE_KERNEL(xor3, "v_xor_b32 %0, %0, %1")
DEFINE_KERNEL(mad_u64_u32, "v_mad_u64_u32 v[10:11], vcc, %0, %1, v[10:11]")
DEFINE_KERNEL(cvt_f32_i32, "v_cvt_f32_i32 %0, %0")
DEFINE_KERNEL(and_or_b32, "v_and_or_b32 %0, %0, %1, %2")
DEFINE_KERNEL(or3_b32, "v_or3_b32 %0, %0, %1, %2")
DEFINE_KERNEL(add3_u32, "v_add3_u32 %0, %0, %1, %2")
DEFINE_KERNEL(bfi_b32, "v_bfi_b32 %0, %0, %1, %2")
DEFINE_KERNEL(ffbl_b32, "v_ffbl_b32 %0, %0")
DEFINE_KERNEL(lshlrev_b32, "v_lshlrev_b32 %0, 8, %0")
DEFINE_KERNEL(lshlrev_b64, "v_lshlrev_b64 v[10:11], %0, v[10:11]")
DEFINE_KERNEL(mbcnt_lo, "v_mbcnt_lo_u32_b32 %0, -1, %0")
DEFINE_KERNEL(cndmask_sgpr, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
DEFINE_KERNEL(cndmask_vcc_set, "v_cndmask_b32_e32 %0, %0, %1, vcc")
DEFINE_KERNEL(cmp_then_cndmask, "v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %2, vcc")
DEFINE_KERNEL(cmp_sgpr_dst, "v_cmp_lt_f32_e64 s[20:21], %0, %1")
DEFINE_KERNEL(cmp_then_4cndmask, "v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %2, vcc\n\tv_cndmask_b32_e32 %0, %0, %1, vcc\n\tv_cndmask_b32_e32 %0, %0, %2, vcc\n\tv_cndmask_b32_e32 %0, %0, %1, vcc")
DEFINE_KERNEL(cndmask_e64_vcc, "v_cndmask_b32_e64 %0, %0, %1, vcc")
DEFINE_KERNEL(smov_vcc_cndmask, "s_mov_b64 vcc, s[20:21]\n\tv_cndmask_b32_e32 %0, %0, %1, vcc")
DEFINE_KERNEL(cmp_sgpr_then_cndmask, "v_cmp_lt_f32_e64 s[20:21], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %2, s[20:21]")
DEFINE_KERNEL(cmp_sgpr_then_4cndmask, "v_cmp_lt_f32_e64 s[20:21], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %2, s[20:21]\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]\n\tv_cndmask_b32_e64 %0, %0, %2, s[20:21]\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]")
DEFINE_KERNEL(add_then_cndmask_sgpr, "v_add_f32 %0, %0, %1\n\tv_cndmask_b32_e64 %0, %0, %2, s[20:21]")
DEFINE_KERNEL(lshlrev_v, "v_lshlrev_b32 %0, %1, %0")
DEFINE_KERNEL(lshrrev_b32, "v_lshrrev_b32 %0, 8, %0")
DEFINE_KERNEL(or_b32, "v_or_b32 %0, %0, %1")
DEFINE_KERNEL(sub_u32, "v_sub_u32 %0, %0, %1")
DEFINE_KERNEL(sub_f32, "v_sub_f32 %0, %0, %1")
DEFINE_KERNEL(fmac_f32, "v_fmac_f32 %0, %1, %2")
DEFINE_KERNEL(mul_f32_lit, "v_mul_f32 %0, 0x34000000, %0")
DEFINE_KERNEL(mul_f32_sgpr, "v_mul_f32 %0, s20, %0")
DEFINE_KERNEL(add_f32_e64_neg, "v_add_f32_e64 %0, -%0, %1")
DEFINE_KERNEL(add_co_u32, "v_add_co_u32 %0, vcc, %0, %1")
DEFINE_KERNEL(cmp_then_2cndmask, "v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %2, vcc\n\tv_cndmask_b32_e32 %0, %0, %1, vcc")
DEFINE_KERNEL(cmp_add_cndmask, "v_cmp_lt_f32 vcc, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %2, vcc")
DEFINE_KERNEL(cmp_2add_cndmask, "v_cmp_lt_f32 vcc, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %2, vcc")
DEFINE_KERNEL(sor_vcc_cndmask, "s_or_b64 vcc, s[20:21], s[20:21]\n\tv_cndmask_b32_e32 %0, %0, %1, vcc")
DEFINE_KERNEL(cndmask_indep_dst, "v_cndmask_b32_e32 v10, %0, %1, vcc")
DEFINE_KERNEL(cndmask_e32_inline, "v_cndmask_b32_e32 %0, 1, %1, vcc")
DEFINE_KERNEL(cmp_then_4cndmask_indep, "v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32_e32 v10, %0, %2, vcc\n\tv_cndmask_b32_e32 v11, %0, %1, vcc\n\tv_cndmask_b32_e32 v10, %0, %2, vcc\n\tv_cndmask_b32_e32 v11, %0, %1, vcc")
DEFINE_KERNEL(cmp_then_4cndmask_e64vcc, "v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32_e64 %0, %0, %2, vcc\n\tv_cndmask_b32_e64 %0, %0, %1, vcc\n\tv_cndmask_b32_e64 %0, %0, %2, vcc\n\tv_cndmask_b32_e64 %0, %0, %1, vcc")
DEFINE_KERNEL(add_4x, "v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2")
DEFINE_KERNEL(add_f32_sgpr, "v_add_f32 %0, s20, %0")
DEFINE_KERNEL(fma_f32_sgpr, "v_fma_f32 %0, %0, s20, %1")
DEFINE_KERNEL(and_b32_sgpr, "v_and_b32 %0, s20, %0")
DEFINE_KERNEL(mul_lo_u32_sgpr, "v_mul_lo_u32 %0, %0, s20")
DEFINE_KERNEL(sub_f32_inline, "v_sub_f32 %0, 1.0, %0")
DEFINE_KERNEL(mul_f32_sdwa_dummy, "v_mul_f32 %0, %0, %0")
DEFINE_KERNEL(salu_vcc_4cndmask, "s_or_b64 vcc, s[20:21], exec\n\ts_nop 1\n\tv_cndmask_b32_e32 v10, %0, %2, vcc\n\tv_cndmask_b32_e32 v11, %0, %1, vcc\n\tv_cndmask_b32_e32 v10, %0, %2, vcc\n\tv_cndmask_b32_e32 v11, %0, %1, vcc")
DEFINE_KERNEL(salu_vcc_1cndmask, "s_or_b64 vcc, s[20:21], exec\n\ts_nop 1\n\tv_cndmask_b32_e32 v10, %0, %2, vcc")
DEFINE_KERNEL(salu_vcc_2cndmask, "s_or_b64 vcc, s[20:21], exec\n\ts_nop 1\n\tv_cndmask_b32_e32 v10, %0, %2, vcc\n\tv_cndmask_b32_e32 v11, %0, %1, vcc")
DEFINE_KERNEL(cmp_nop_2cndmask, "v_cmp_lt_f32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e32 v10, %0, %2, vcc\n\tv_cndmask_b32_e32 v11, %0, %1, vcc")
DEFINE_KERNEL(cmp_nop_cnd_add_cnd, "v_cmp_lt_f32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e32 v10, %0, %2, vcc\n\tv_add_f32 %0, %0, %1\n\tv_cndmask_b32_e32 v11, %0, %1, vcc")
DEFINE_KERNEL(cmp_nop_cnd_2add_cnd, "v_cmp_lt_f32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e32 v10, %0, %2, vcc\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\tv_cndmask_b32_e32 v11, %0, %1, vcc")
DEFINE_KERNEL(cmp_nop_4cndmask_e64, "v_cmp_lt_f32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e64 v10, %0, %2, vcc\n\tv_cndmask_b32_e64 v11, %0, %1, vcc\n\tv_cndmask_b32_e64 v10, %0, %2, vcc\n\tv_cndmask_b32_e64 v11, %0, %1, vcc")
DEFINE_KERNEL(exec_masked_4mov, "s_and_saveexec_b64 s[20:21], vcc\n\tv_mov_b32 v10, %1\n\tv_mov_b32 v11, %2\n\tv_mov_b32 v10, %1\n\tv_mov_b32 v11, %2\n\ts_or_b64 exec, exec, s[20:21]")
DEFINE_KERNEL(min_u32, "v_min_u32 %0, %0, %1")
DEFINE_KERNEL(sub_f32_dpp, "v_sub_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
DEFINE_KERNEL(mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
DEFINE_KERNEL(readlane, "v_readlane_b32 s20, %0, 5")
DEFINE_KERNEL(ds_bpermute, "ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)")
DEFINE_KERNEL(ds_bpermute_nowait, "ds_bpermute_b32 %0, %1, %2")

// packed ops need register pairs
#define DEFINE_PK_KERNEL(NAME, ASMSTR)                                                         \
    __global__ __launch_bounds__(256) void k_##NAME(float* out, int iters, float seed)          \
    {                                                                                          \
        typedef float float2v __attribute__((ext_vector_type(2)));                             \
        float2v r[8];                                                                          \
        for (int i = 0; i < 8; ++i) { r[i].x = seed + threadIdx.x * 1e-3f + i; r[i].y = r[i].x + 1; } \
        float2v a = {seed * 0.5f + 1.0f, seed}, b = {seed + 3.0f, seed};                       \
        for (int it = 0; it < iters; ++it) {                                                   \
            _Pragma("unroll") for (int u = 0; u < kUnroll / 8; ++u) {                          \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                \
                    asm volatile(ASMSTR : "+v"(r[i]) : "v"(a), "v"(b));                        \
                }                                                                              \
            }                                                                                  \
        }                                                                                      \
        float s = 0;                                                                           \
        for (int i = 0; i < 8; ++i) s += r[i].x + r[i].y;                                      \
        if (s == 12345.678f) out[0] = s;                                                       \
    }
DEFINE_PK_KERNEL(pk_add_f32, "v_pk_add_f32 %0, %0, %1")
DEFINE_PK_KERNEL(pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
DEFINE_PK_KERNEL(pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %2")

// exec-mask experiments: does a half-empty wave issue faster?
#define DEFINE_MASKED_KERNEL(NAME, COND)                                                       \
    __global__ __launch_bounds__(256) void k_##NAME(float* out, int iters, float seed)          \
    {                                                                                          \
        float r[8];                                                                            \
        for (int i = 0; i < 8; ++i) r[i] = seed + threadIdx.x * 1e-3f + i;                     \
        float a = seed * 0.5f + 1.0f;                                                          \
        const unsigned lane = threadIdx.x & 63u;                                               \
        if (COND) {                                                                            \
            for (int it = 0; it < iters; ++it) {                                               \
                _Pragma("unroll") for (int u = 0; u < kUnroll / 8; ++u) {                      \
                    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                            \
                        asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));            \
                    }                                                                          \
                }                                                                              \
            }                                                                                  \
        }                                                                                      \
        float s = 0;                                                                           \
        for (int i = 0; i < 8; ++i) s += r[i];                                                 \
        if (s == 12345.678f) out[0] = s;                                                       \
    }
DEFINE_MASKED_KERNEL(add_low32, lane < 32u)
DEFINE_MASKED_KERNEL(add_high32, lane >= 32u)
DEFINE_MASKED_KERNEL(add_even, (lane & 1u) == 0u)
DEFINE_MASKED_KERNEL(add_one_lane, lane == 5u)
DEFINE_MASKED_KERNEL(add_low16, lane < 16u)

// dependent chain (latency-bound): one register
__global__ __launch_bounds__(256) void k_dep_add(float* out, int iters, float seed)
{
    float r = seed + threadIdx.x;
    float a = seed + 1.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(r) : "v"(a));
        }
    }
    if (r == 12345.678f) out[0] = r;
}
__global__ __launch_bounds__(256) void k_dep_rcp(float* out, int iters, float seed)
{
    float r = seed + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            asm volatile("v_rcp_f32 %0, %0" : "+v"(r));
        }
    }
    if (r == 12345.678f) out[0] = r;
}
// LDS broadcast read
__global__ __launch_bounds__(256) void k_ds_read_b128_bcast(float* out, int iters, float seed)
{
    __shared__ float4 buf[64];
    if (threadIdx.x < 64) buf[threadIdx.x] = make_float4(seed, seed, seed, seed);
    __syncthreads();
    float4 acc = make_float4(0, 0, 0, 0);
    int idx = (int)seed & 63;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            float4 v;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(((idx + u) & 63) * 16)) : "memory");
            acc.x += v.x;
        }
    }
    if (acc.x == 12345.678f) out[0] = acc.x;
}

typedef void (*kernel_t)(float*, int, float);
struct Entry { const char* name; kernel_t fn; };

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    double clock_hz = prop.clockRate * 1e3;
    printf("device %s, %d CUs, clockRate %.0f MHz\n", prop.name, cus, clock_hz / 1e6);
    float* out;
    CHECK(hipMalloc(&out, 4));
    std::vector<Entry> entries = {
#define E(n) { #n, k_##n }
        E(add_f32), E(add_low32), E(add_high32), E(add_even), E(add_one_lane), E(add_low16), E(mul_f32), E(fma_f32), E(pk_add_f32), E(pk_mul_f32), E(pk_fma_f32), E(mul_lo_u32), E(mul_hi_u32),
        E(mul_u32_u24), E(mad_u32_u24), E(mad_u64_u32), E(add_u32), E(and_b32), E(xor3), E(lshl_add_u32), E(bfe_i32),
        E(cvt_f32_u32), E(cvt_u32_f32), E(rcp_f32), E(sqrt_f32), E(rsq_f32), E(div_scale), E(div_fmas), E(div_fixup),
        E(cndmask), E(cndmask_sgpr), E(cndmask_vcc_set), E(cmp_then_cndmask), E(cmp_sgpr_dst), E(cvt_f32_i32), E(and_or_b32), E(or3_b32), E(add3_u32), E(bfi_b32), E(ffbl_b32), E(lshlrev_b32), E(lshlrev_b64), E(mbcnt_lo), E(cmp_then_4cndmask), E(cndmask_e64_vcc), E(smov_vcc_cndmask), E(cmp_sgpr_then_cndmask), E(cmp_sgpr_then_4cndmask), E(add_then_cndmask_sgpr), E(lshlrev_v), E(lshrrev_b32), E(or_b32), E(sub_u32), E(sub_f32), E(fmac_f32), E(mul_f32_lit), E(mul_f32_sgpr), E(add_f32_e64_neg), E(add_co_u32), E(cmp_then_2cndmask), E(cmp_add_cndmask), E(cmp_2add_cndmask), E(sor_vcc_cndmask), E(cndmask_indep_dst), E(cndmask_e32_inline), E(cmp_then_4cndmask_indep), E(cmp_then_4cndmask_e64vcc), E(add_4x), E(add_f32_sgpr), E(fma_f32_sgpr), E(and_b32_sgpr), E(mul_lo_u32_sgpr), E(sub_f32_inline), E(mul_f32_sdwa_dummy), E(salu_vcc_4cndmask), E(salu_vcc_1cndmask), E(salu_vcc_2cndmask), E(cmp_nop_2cndmask), E(cmp_nop_cnd_add_cnd), E(cmp_nop_cnd_2add_cnd), E(cmp_nop_4cndmask_e64), E(exec_masked_4mov), E(min_u32), E(sub_f32_dpp), E(mov_dpp), E(readlane), E(ds_bpermute), E(ds_bpermute_nowait), E(cmp_lt), E(floor_f32), E(ldexp_f32), E(max_f32), E(mov), E(dep_add), E(dep_rcp), E(ds_read_b128_bcast),
    };
    const int iters = 2000;
    for (int waves_per_simd : {4}) {
        printf("---- %d wave(s) per SIMD ----\n", waves_per_simd);
        int blocks = cus * waves_per_simd; // 256 threads = 4 waves = 1 per SIMD
        for (auto& e : entries) {
            hipEvent_t a, b;
            CHECK(hipEventCreate(&a));
            CHECK(hipEventCreate(&b));
            e.fn<<<blocks, 256>>>(out, 10, 1.5f); // warm-up
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(a));
            e.fn<<<blocks, 256>>>(out, iters, 1.5f);
            CHECK(hipGetLastError());
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms;
            CHECK(hipEventElapsedTime(&ms, a, b));
            double instr_per_simd = (double)iters * kUnroll * waves_per_simd;
            double cycles = ms * 1e-3 * clock_hz;
            printf("%-22s %8.3f ms  %6.2f cycles/instr/SIMD\n", e.name, ms, cycles / instr_per_simd);
        }
    }
    return 0;
}

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

#define DEFINE_KERNEL(NAME, ASMSTR)                                                            \
    __global__ __launch_bounds__(256) void k_##NAME(float* out, int iters, float seed)          \
    {                                                                                          \
        float r[8];                                                                            \
        for (int i = 0; i < 8; ++i) r[i] = seed + threadIdx.x * 1e-3f + i;                     \
        float a = seed * 0.5f + 1.0f, b = seed + 3.0f;                                         \
        for (int it = 0; it < iters; ++it) {                                                   \
            _Pragma("unroll") for (int u = 0; u < kUnroll / 8; ++u) {                          \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                \
                    asm volatile(ASMSTR : "+v"(r[i]) : "v"(a), "v"(b));                        \
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
        E(cndmask), E(cmp_lt), E(floor_f32), E(ldexp_f32), E(max_f32), E(mov), E(dep_add), E(dep_rcp), E(ds_read_b128_bcast),
    };
    const int iters = 2000;
    for (int waves_per_simd : {1, 2, 4, 8}) {
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

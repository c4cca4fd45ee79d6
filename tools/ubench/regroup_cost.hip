// Micro-benchmark: what regrouping rays across the lanes of a workgroup through LDS costs on gfx950.
// Every lane holds a ray's state (19 words, as the trace kernel's lanes do); one "regroup" = every lane
// writes its state to a queue slot in LDS (slot = a compaction index from a ballot/prefix count, scattered
// over the workgroup), the workgroup synchronises, and every lane reads the state of the slot it now owns.
// Prints the time per ray and regroup at the trace kernel's occupancy (8 waves per SIMD).  Diagnostic only.
//   hipcc --offload-arch=gfx950 -O2 -o regroup_cost regroup_cost.hip && ./regroup_cost
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kWords = 20; // 19 words of state, padded to five 16-byte rows
constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void regroup(float* out, int iters, float seed, int do_regroup)
{
    __shared__ float4 queue[kBlock * kWords / 4];
    float s[kWords];
    for (int i = 0; i < kWords; ++i) {
        s[i] = seed + threadIdx.x * 1e-3f + i;
    }
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (int it = 0; it < iters; ++it) {
        // a little arithmetic per segment so that the state changes (and a data-dependent predicate)
        for (int i = 0; i < kWords; ++i) {
            s[i] = s[i] * 1.0001f + 0.5f;
        }
        if (do_regroup) {
            const bool key = (__float_as_uint(s[3]) >> 3) & 1u; // which queue the ray goes to
            const unsigned long long m = __builtin_amdgcn_ballot_w64(key);
            const unsigned prefix = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            const unsigned ones = (unsigned)__popcll(m);
            // key lanes fill the wave's region from the front, the others from the back; regions rotate over the waves
            const unsigned slot_in_wave = key ? prefix : 63u - (lane - prefix);
            const unsigned slot = (((wave + (unsigned)it) & 3u) << 6) + slot_in_wave + 0u * ones;
            float4* dst = queue + slot * (kWords / 4);
            for (int r = 0; r < kWords / 4; ++r) {
                dst[r] = make_float4(s[4 * r], s[4 * r + 1], s[4 * r + 2], s[4 * r + 3]);
            }
            __syncthreads();
            const float4* src = queue + threadIdx.x * (kWords / 4);
            for (int r = 0; r < kWords / 4; ++r) {
                const float4 v = src[r];
                s[4 * r] = v.x; s[4 * r + 1] = v.y; s[4 * r + 2] = v.z; s[4 * r + 3] = v.w;
            }
            __syncthreads();
        }
    }
    float acc = 0.0f;
    for (int i = 0; i < kWords; ++i) {
        acc += s[i];
    }
    if (acc == 12345.678f) {
        out[0] = acc;
    }
}

int main()
{
    float* out = nullptr;
    if (hipMalloc(&out, 64) != hipSuccess) {
        return 1;
    }
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    const int iters = 2000;
    const int blocks = 256 * 8; // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    double base = 0.0;
    for (int mode = 0; mode < 2; ++mode) {
        float ms = 0.0f;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(regroup, dim3(blocks), dim3(kBlock), 0, 0, out, iters, 1.0f, mode);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            (void)hipEventElapsedTime(&ms, a, b);
        }
        const double rays = (double)blocks * kBlock * iters;
        printf("%s: %.3f ms (%.2e ray-steps/s)\n", mode ? "with regroup" : "arithmetic only", ms, rays / (ms * 1e-3));
        if (mode == 0) {
            base = ms;
        } else {
            printf("one regroup of a 19-word ray state through LDS: %.2e per second per GPU; the benchmark's 4.13e9 segments x 3 "
                   "regroups would take %.1f ms (the whole trace kernel takes 28.6 ms)\n",
                   rays / ((ms - base) * 1e-3), 4.13e9 * 3.0 / (rays / ((ms - base) * 1e-3)) * 1e3);
        }
    }
    return 0;
}

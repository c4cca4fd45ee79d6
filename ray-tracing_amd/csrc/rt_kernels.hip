// HIP kernels of the sphere path tracer for gfx950 (MI355X).
//
// rt_trace_kernel is the counterpart of main() in shaders/raytrace.comp.glsl:298-343,
// restructured for 64-wide wavefronts:
//   * one lane owns one pixel for the whole launch and walks its frames and the
//     samples of each frame in order, so the per-(pixel,frame) random stream of
//     :310-331 is consumed exactly as the shader consumes it;
//   * the bounce loop (:242-293) and the sample loop (:315-331) are flattened into
//     one loop whose iteration is "one ray segment": a lane whose path has ended
//     starts its next camera ray in the same iteration instead of idling until the
//     longest path of the wave has finished;
//   * the sphere list is staged once per workgroup in LDS as (centre, radius^2)
//     float4 records that every lane reads at the same address (broadcast), and the
//     material records are fetched only for the winning sphere;
//   * the accumulator texel is read once and written once per launch as one float4.
//
// Compile with -ffp-contract=off (see rt_device.hpp).
#include "rt_device.hpp"
#include "rt_kernels.hpp"

namespace rt {

namespace {

constexpr int kBlockThreads = 256;
constexpr float kTMin = 0.001f;

// Copies the scene into LDS: geometry always, material records when they fit.
template <bool MAT_IN_LDS>
__device__ __forceinline__ void stage_scene(const rt_gpu_sphere* __restrict__ spheres, uint32_t count,
                                            float4* geom, float4* mat0, float4* mat1)
{
    for (uint32_t i = threadIdx.x; i < count; i += kBlockThreads) {
        const float4* src = reinterpret_cast<const float4*>(spheres + i);
        float4 cr = src[0];
        geom[i] = make_float4(cr.x, cr.y, cr.z, cr.w * cr.w);
        if (MAT_IN_LDS) {
            float4 al = src[1];
            mat0[i] = make_float4(al.x, al.y, al.z, cr.w);
            mat1[i] = src[2];
        }
    }
    __syncthreads();
}

template <bool MAT_IN_LDS>
__global__ __launch_bounds__(kBlockThreads) void rt_trace_kernel(TraceArgs A)
{
    extern __shared__ float4 lds[];
    const uint32_t sphere_count = A.params.frame_sample_depth_count[3];
    float4* geom = lds;
    float4* mat0 = lds + sphere_count;
    float4* mat1 = lds + 2 * sphere_count;
    stage_scene<MAT_IN_LDS>(A.spheres, sphere_count, geom, mat0, mat1);

    // One wave = one 8x8 pixel tile (the reference's workgroup shape, :5).
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t tile = blockIdx.x * (kBlockThreads / 64) + (threadIdx.x >> 6);
    const uint32_t tiles_x = (A.width + 7u) >> 3;
    const uint32_t tile_y = tile / tiles_x;
    const uint32_t tile_x = tile - tile_y * tiles_x;
    const uint32_t px = tile_x * 8u + (lane & 7u);
    const uint32_t py = A.row_begin + tile_y * 8u + (lane >> 3);

    // :302-305 bounds check (plus the row band of this shard)
    if (px >= A.width || py >= A.row_end) {
        return;
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
    const float fpx = (float)(int32_t)px;
    // float(params.resolution.y - 1u - pixel.y), :320
    const float fpy_flipped = (A.params.resolution[1] - 1.0f) - (float)(int32_t)py;

    const size_t texel = (size_t)py * A.width + px;
    const float4 prev = A.accum[texel]; // :333
    f3 total = mk(prev.x, prev.y, prev.z);

    uint32_t frame = A.first_frame;
    uint32_t frames_left = A.n_frames;
    uint32_t samples_left = spp;
    uint32_t rng = hash3(px, py, frame); // :310-311
    f3 frame_sum = mk(0.0f, 0.0f, 0.0f); // `accum` of :313
    bool need_ray = true;
    uint32_t depth = 0;
    f3 ro = origin, rd = origin;
    f3 throughput = mk(1.0f, 1.0f, 1.0f);

    while (frames_left != 0u) {
        if (need_ray) {
            if (samples_left == 0u) {
                // end of one dispatch for this pixel: :333-335
                total = total + frame_sum;
                frame_sum = mk(0.0f, 0.0f, 0.0f);
                --frames_left;
                if (frames_left == 0u) {
                    break;
                }
                frame += A.frame_stride;
                rng = hash3(px, py, frame);
                samples_left = spp;
            }
            --samples_left;
            // camera ray, :317-328
            float u_coord = (fpx + rnd(rng)) * inv_w;
            float v_coord = (fpy_flipped + rnd(rng)) * inv_h;
            float dx, dy;
            random_in_unit_disk(rng, dx, dy);
            float rdx = lens_radius * dx;
            float rdy = lens_radius * dy;
            f3 offset = cam_u * rdx + cam_v * rdy;
            f3 dir = lower_left + horizontal * u_coord + vertical * v_coord - origin - offset;
            ro = origin + offset;
            rd = normalize(dir);
            throughput = mk(1.0f, 1.0f, 1.0f);
            depth = 0u;
            need_ray = false;
        }

        // closest hit over all spheres, :244-259
        float closest = 1e20f;
        int best = -1;
        const float a = dot(rd, rd);
        for (uint32_t i = 0; i < sphere_count; ++i) {
            float root;
            if (sphere_root(geom[i], ro, rd, a, kTMin, closest, root)) {
                closest = root;
                best = (int)i;
            }
        }

        if (best < 0) {
            // sky, :261-269
            f3 unit_dir = normalize(rd);
            float t = 0.5f * (unit_dir.y + 1.0f);
            float omt = 1.0f - t;
            f3 sky = mk(omt + 0.5f * t, omt + 0.7f * t, omt + 1.0f * t);
            f3 radiance = mk(0.0f, 0.0f, 0.0f) + throughput * sky;
            frame_sum = frame_sum + radiance; // :330
            need_ray = true;
            continue;
        }

        // winner's record, :135-149
        float4 g = geom[best];
        float4 m0, m1;
        if (MAT_IN_LDS) {
            m0 = mat0[best];
            m1 = mat1[best];
        } else {
            const float4* src = reinterpret_cast<const float4*>(A.spheres + best);
            float4 cr = src[0];
            float4 al = src[1];
            m0 = make_float4(al.x, al.y, al.z, cr.w);
            m1 = src[2];
        }
        f3 p, n;
        bool front;
        hit_finish(mk(g.x, g.y, g.z), m0.w, ro, rd, closest, p, n, front);
        const uint32_t mat = (uint32_t)(m1.x + 0.5f);
        f3 albedo = mk(m0.x, m0.y, m0.z);
        bool scattered = true;

        if (mat == 0u) {
            // scatterLambert, :155-174
            f3 ruv = normalize(random_in_unit_sphere(rng));
            f3 target = p + n + ruv;
            ro = p + n * 0.001f;
            rd = normalize(target - p);
            if (m1.w > 0.5f) {
                float checker_index = floorf(p.x) + floorf(p.z);
                float checker = checker_index - 2.0f * floorf(checker_index / 2.0f);
                float tone = (checker < 1.0f) ? 0.05f : 0.95f;
                albedo = albedo * tone;
            }
            throughput = throughput * albedo;
        } else if (mat == 1u) {
            // scatterMetal, :199-208
            f3 reflected = reflect(normalize(rd), n);
            ro = p + n * 0.001f;
            rd = reflected + random_in_unit_sphere(rng) * m1.y;
            throughput = throughput * albedo;
            scattered = dot(rd, n) > 0.0f;
        } else {
            // scatterDielectric, :210-234
            throughput = throughput * mk(1.0f, 1.0f, 1.0f);
            float eta_ratio = front ? (1.0f / m1.z) : m1.z;
            f3 unit_dir = normalize(rd);
            float cos_theta = glsl_min(dot(-unit_dir, n), 1.0f);
            float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
            bool do_reflect = eta_ratio * sin_theta > 1.0f;
            if (!do_reflect) {
                // `||` short-circuits: the draw happens only on this side
                float sch = schlick(cos_theta, eta_ratio);
                do_reflect = sch > rnd(rng);
            }
            f3 direction;
            if (do_reflect || !refract_vec(unit_dir, n, eta_ratio, direction)) {
                direction = reflect(unit_dir, n);
            }
            ro = p + direction * 0.001f;
            rd = direction;
        }

        ++depth;
        if (!scattered || depth >= max_depth) {
            need_ray = true; // :287-290 break, or the bounce budget of :242 is spent
        }
    }

    A.accum[texel] = make_float4(total.x, total.y, total.z, 1.0f); // :335
    if (A.rng_out != nullptr) {
        A.rng_out[texel] = rng;
    }
    if (A.output != nullptr) {
        // :337 (frame + 1u) * spp wraps in uint32 before the conversion
        float sample_count = (float)((frame + 1u) * spp_raw);
        A.output[texel] = tonemap_rgba8(total, sample_count);
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
    bool hit = sphere_root(geom, o, d, dot(d, d), tmin, tmax, root);
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

size_t trace_lds_bytes(uint32_t sphere_count, bool* mat_in_lds)
{
    // Up to 1024 spheres the whole 48-byte record lives in LDS (48 KiB, three
    // workgroups per CU); beyond that only the 16-byte geometry record does.
    const bool full = sphere_count <= kMaxSpheresFullLds;
    if (mat_in_lds != nullptr) {
        *mat_in_lds = full;
    }
    return (size_t)sphere_count * (full ? 48u : 16u);
}

hipError_t launch_trace(const TraceArgs& args, hipStream_t stream)
{
    const uint32_t rows = args.row_end - args.row_begin;
    const uint32_t tiles_x = (args.width + 7u) / 8u;
    const uint32_t tiles_y = (rows + 7u) / 8u;
    const uint64_t tiles = (uint64_t)tiles_x * tiles_y;
    const uint32_t waves_per_block = kBlockThreads / 64;
    const uint64_t blocks = (tiles + waves_per_block - 1) / waves_per_block;
    if (blocks == 0 || args.n_frames == 0) {
        return hipSuccess;
    }
    bool full = false;
    const size_t lds = trace_lds_bytes(args.params.frame_sample_depth_count[3], &full);
    if (full) {
        hipLaunchKernelGGL(rt_trace_kernel<true>, dim3((uint32_t)blocks), dim3(kBlockThreads), lds, stream, args);
    } else {
        hipLaunchKernelGGL(rt_trace_kernel<false>, dim3((uint32_t)blocks), dim3(kBlockThreads), lds, stream, args);
    }
    return hipGetLastError();
}

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

hipError_t launch_probe_rng(const uint32_t* xyz, uint32_t n, uint32_t draws, uint32_t* out, hipStream_t stream)
{
    hipLaunchKernelGGL(rt_probe_rng_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, xyz, n, draws, out);
    return hipGetLastError();
}

} // namespace rt

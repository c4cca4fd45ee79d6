// The multi-GPU leg of the MI355X sphere path tracer: the C ABI of include/rt_mi355x_rccl.h.
//
// One process per GPU.  A step's image is shared out over the ranks either as interleaved stripes of rows
// (rt_render_stripes; put together by one ncclAllGather of equal-sized packed shards) or by frame index
// (rt_render_shard with a frame stride; the accumulators are summed by ncclAllReduce and alpha is set back to
// the 1 that raytrace.comp.glsl:335 stores with every frame).  The reference has one device
// (src/vk/VulkanContext.cpp:413-414) and no counterpart for any of this.
//
// The kernels here are plain HBM-bound copies, one float4 per thread, rows along blockIdx.y so that no thread
// divides: pack (this rank's stripes of the accumulator -> a dense send buffer), unpack (the gathered buffer,
// rank-major -> image order) and the alpha fix.  Over fully connected xGMI each rank sends its 1/G of the image
// to seven peers on seven links; the collective is RCCL's to schedule.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <new>
#include <string>

#include "../../include/rt_mi355x_rccl.h"

struct rt_comm {
    ncclComm_t comm = nullptr;
    bool owned = false;
    int world = 1;
    int rank = 0;
    int device = 0;
    void* staging[2] = { nullptr, nullptr }; // packed stripes, or a copy of the accumulator
    size_t staging_bytes[2] = { 0, 0 };
    void* gathered = nullptr;                // the ranks' packed stripes, rank-major
    size_t gathered_bytes = 0;
    std::string error;
};

namespace {

thread_local std::string g_comm_error;
constexpr int kThreads = 256;

int fail(rt_comm* c, int status, const std::string& text)
{
    (c != nullptr ? c->error : g_comm_error) = text;
    return status;
}

int fail_hip(rt_comm* c, hipError_t e, const char* expr, int line)
{
    char buf[400];
    std::snprintf(buf, sizeof(buf), "hipError(%d: %s) for %s at %s:%d", (int)e, hipGetErrorName(e), expr, __FILE__, line);
    (void)hipGetLastError();
    return fail(c, e == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP, buf);
}

int fail_nccl(rt_comm* c, ncclResult_t r, const char* expr, int line)
{
    char buf[400];
    std::snprintf(buf, sizeof(buf), "ncclResult(%d: %s) for %s at %s:%d", (int)r, ncclGetErrorString(r), expr, __FILE__, line);
    return fail(c, RT_ERR_HIP, buf);
}

#define RC_HIP(c, expr)                                      \
    do {                                                     \
        hipError_t rc_e_ = (expr);                           \
        if (rc_e_ != hipSuccess) {                           \
            return fail_hip((c), rc_e_, #expr, __LINE__);    \
        }                                                    \
    } while (0)

#define RC_NCCL(c, expr)                                     \
    do {                                                     \
        ncclResult_t rc_r_ = (expr);                         \
        if (rc_r_ != ncclSuccess) {                          \
            return fail_nccl((c), rc_r_, #expr, __LINE__);   \
        }                                                    \
    } while (0)

// Makes the communicator's device current for one call.
class DeviceScope {
public:
    explicit DeviceScope(int device)
    {
        if (hipGetDevice(&previous_) != hipSuccess) {
            previous_ = -1;
        }
        if (previous_ != device) {
            status_ = hipSetDevice(device);
            switched_ = status_ == hipSuccess;
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

// Geometry of the stripe split of a width x height image.
struct StripeLayout {
    uint32_t width, height, stripe_rows;
    uint32_t world, rank;
    uint32_t slots;        // stripes per rank, the last one possibly unused or cut short: ceil(stripes / world)
    size_t packed_texels;  // slots * stripe_rows * width
};

StripeLayout stripe_layout(uint32_t width, uint32_t height, uint32_t stripe_rows, int world, int rank)
{
    StripeLayout l;
    l.width = width;
    l.height = height;
    l.stripe_rows = std::min(stripe_rows, height); // as rt_render_stripes clamps it
    l.world = static_cast<uint32_t>(world);
    l.rank = static_cast<uint32_t>(rank);
    const uint32_t stripes = (height + l.stripe_rows - 1u) / l.stripe_rows;
    l.slots = (stripes + l.world - 1u) / l.world;
    l.packed_texels = static_cast<size_t>(l.slots) * l.stripe_rows * width;
    return l;
}

// send[k][within][x] = accum[(rank + k * world) * stripe_rows + within][x], zero below the image.
__global__ __launch_bounds__(kThreads) void rt_pack_stripes_kernel(const float4* __restrict__ accum, float4* __restrict__ send, StripeLayout l)
{
    const uint32_t x = blockIdx.x * kThreads + threadIdx.x;
    if (x >= l.width) {
        return;
    }
    const uint32_t packed_rows = l.slots * l.stripe_rows;
    for (uint32_t row = blockIdx.y; row < packed_rows; row += gridDim.y) {
        const uint32_t k = row / l.stripe_rows; // wave-uniform
        const uint32_t within = row - k * l.stripe_rows;
        const uint64_t y = static_cast<uint64_t>(l.rank + k * l.world) * l.stripe_rows + within;
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (y < l.height) {
            v = accum[y * l.width + x];
        }
        send[static_cast<size_t>(row) * l.width + x] = v;
    }
}

// image[y][x] = recv[rank of the stripe][its slot][within][x]
__global__ __launch_bounds__(kThreads) void rt_unpack_stripes_kernel(const float4* __restrict__ recv, float4* __restrict__ image, StripeLayout l)
{
    const uint32_t x = blockIdx.x * kThreads + threadIdx.x;
    if (x >= l.width) {
        return;
    }
    for (uint32_t y = blockIdx.y; y < l.height; y += gridDim.y) {
        const uint32_t stripe = y / l.stripe_rows; // wave-uniform
        const uint32_t within = y - stripe * l.stripe_rows;
        const uint32_t owner = stripe % l.world;
        const uint32_t k = stripe / l.world;
        const size_t row = (static_cast<size_t>(owner) * l.slots + k) * l.stripe_rows + within;
        image[static_cast<size_t>(y) * l.width + x] = recv[row * l.width + x];
    }
}

// raytrace.comp.glsl:335 stores alpha = 1 with every frame; a sum over G ranks holds G.
__global__ __launch_bounds__(kThreads) void rt_alpha_one_kernel(float4* __restrict__ image, size_t texels)
{
    const size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x;
    if (i < texels) {
        image[i].w = 1.0f;
    }
}

dim3 row_grid(uint32_t width, uint32_t rows)
{
    return dim3((width + kThreads - 1u) / kThreads, std::min<uint32_t>(std::max<uint32_t>(rows, 1u), 32768u), 1u);
}

int ensure(rt_comm* c, void*& buffer, size_t& have, size_t need)
{
    if (need <= have) {
        return RT_OK;
    }
    if (buffer != nullptr) {
        RC_HIP(c, hipFree(buffer)); // waits for the work that uses it
        buffer = nullptr;
        have = 0;
    }
    RC_HIP(c, hipMalloc(&buffer, need));
    have = need;
    return RT_OK;
}

struct Extent {
    uint32_t width = 0, height = 0;
    float4* accum = nullptr;
};

int image_of(rt_comm* c, rt_handle h, Extent& e)
{
    if (h == nullptr || rt_get_extent(h, &e.width, &e.height) != RT_OK) {
        return fail(c, RT_ERR_BAD_ARGUMENT, "bad argument: tracer handle");
    }
    e.accum = static_cast<float4*>(rt_accum_device_ptr(h));
    if (e.accum == nullptr) {
        return fail(c, RT_ERR_NO_DEVICE, "the tracer handle has no accumulator on a device");
    }
    return RT_OK;
}

int new_comm(ncclComm_t comm, bool owned, int world, int rank, int device_id, rt_comm_handle* out)
{
    rt_comm* c = new (std::nothrow) rt_comm();
    if (c == nullptr) {
        if (owned) {
            (void)ncclCommDestroy(comm);
        }
        return fail(nullptr, RT_ERR_OUT_OF_MEMORY, "host allocation failed");
    }
    c->comm = comm;
    c->owned = owned;
    c->world = world;
    c->rank = rank;
    c->device = device_id;
    *out = c;
    return RT_OK;
}

} // namespace

extern "C" {

int rt_comm_unique_id(unsigned char id[RT_COMM_ID_BYTES])
{
    static_assert(sizeof(ncclUniqueId) == RT_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    if (id == nullptr) {
        return fail(nullptr, RT_ERR_BAD_ARGUMENT, "bad argument: id is NULL");
    }
    ncclUniqueId u;
    RC_NCCL(nullptr, ncclGetUniqueId(&u));
    std::copy(u.internal, u.internal + RT_COMM_ID_BYTES, reinterpret_cast<char*>(id));
    return RT_OK;
}

int rt_comm_create(const unsigned char id[RT_COMM_ID_BYTES], int world, int rank, int device_id, rt_comm_handle* out)
{
    if (out == nullptr || id == nullptr || world < 1 || rank < 0 || rank >= world) {
        return fail(nullptr, RT_ERR_BAD_ARGUMENT, "bad argument: id/out is NULL or rank outside the world");
    }
    *out = nullptr;
    DeviceScope scope(device_id);
    RC_HIP(nullptr, scope.status());
    ncclUniqueId u;
    std::copy(id, id + RT_COMM_ID_BYTES, reinterpret_cast<unsigned char*>(u.internal));
    ncclComm_t comm = nullptr;
    RC_NCCL(nullptr, ncclCommInitRank(&comm, world, u, rank));
    return new_comm(comm, true, world, rank, device_id, out);
}

int rt_comm_adopt(void* nccl_comm, int world, int rank, int device_id, rt_comm_handle* out)
{
    if (out == nullptr || nccl_comm == nullptr || world < 1 || rank < 0 || rank >= world) {
        return fail(nullptr, RT_ERR_BAD_ARGUMENT, "bad argument: communicator/out is NULL or rank outside the world");
    }
    *out = nullptr;
    return new_comm(static_cast<ncclComm_t>(nccl_comm), false, world, rank, device_id, out);
}

int rt_comm_destroy(rt_comm_handle c)
{
    if (c == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    {
        DeviceScope scope(c->device);
        for (void* p : { c->staging[0], c->staging[1], c->gathered }) {
            if (p != nullptr) {
                (void)hipFree(p);
            }
        }
        if (c->owned && c->comm != nullptr) {
            (void)ncclCommDestroy(c->comm);
        }
        (void)hipGetLastError();
    }
    delete c;
    return RT_OK;
}

int rt_comm_world(rt_comm_handle c) { return c != nullptr ? c->world : 0; }
int rt_comm_rank(rt_comm_handle c) { return c != nullptr ? c->rank : -1; }

int rt_comm_pack_stripes(rt_handle h, rt_comm_handle c, uint32_t stripe_rows, uint32_t slot, void* stream)
{
    if (c == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    if (slot > 1u || stripe_rows == 0u) {
        return fail(c, RT_ERR_BAD_ARGUMENT, "bad argument: slot must be 0 or 1 and stripe_rows >= 1");
    }
    Extent e;
    int rc = image_of(c, h, e);
    if (rc != RT_OK) {
        return rc;
    }
    DeviceScope scope(c->device);
    RC_HIP(c, scope.status());
    const StripeLayout l = stripe_layout(e.width, e.height, stripe_rows, c->world, c->rank);
    rc = ensure(c, c->staging[slot], c->staging_bytes[slot], l.packed_texels * sizeof(float4));
    if (rc != RT_OK) {
        return rc;
    }
    hipLaunchKernelGGL(rt_pack_stripes_kernel, row_grid(l.width, l.slots * l.stripe_rows), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream), e.accum, static_cast<float4*>(c->staging[slot]), l);
    RC_HIP(c, hipGetLastError());
    return RT_OK;
}

int rt_comm_gather_packed(rt_handle h, rt_comm_handle c, uint32_t stripe_rows, uint32_t slot, void* dst_image, void* stream)
{
    if (c == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    if (slot > 1u || stripe_rows == 0u || dst_image == nullptr) {
        return fail(c, RT_ERR_BAD_ARGUMENT, "bad argument: slot must be 0 or 1, stripe_rows >= 1, dst_image not NULL");
    }
    Extent e;
    int rc = image_of(c, h, e);
    if (rc != RT_OK) {
        return rc;
    }
    DeviceScope scope(c->device);
    RC_HIP(c, scope.status());
    const StripeLayout l = stripe_layout(e.width, e.height, stripe_rows, c->world, c->rank);
    if (c->staging[slot] == nullptr || c->staging_bytes[slot] < l.packed_texels * sizeof(float4)) {
        return fail(c, RT_ERR_BAD_ARGUMENT, "bad argument: nothing was packed into this slot for this image and stripe height");
    }
    rc = ensure(c, c->gathered, c->gathered_bytes, l.packed_texels * sizeof(float4) * l.world);
    if (rc != RT_OK) {
        return rc;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    RC_NCCL(c, ncclAllGather(c->staging[slot], c->gathered, l.packed_texels * 4u, ncclFloat, c->comm, s));
    hipLaunchKernelGGL(rt_unpack_stripes_kernel, row_grid(l.width, l.height), dim3(kThreads), 0, s,
                       static_cast<const float4*>(c->gathered), static_cast<float4*>(dst_image), l);
    RC_HIP(c, hipGetLastError());
    return RT_OK;
}

int rt_gather_stripes(rt_handle h, rt_comm_handle c, uint32_t stripe_rows, void* dst_image, void* stream)
{
    const int rc = rt_comm_pack_stripes(h, c, stripe_rows, 0u, stream);
    return rc != RT_OK ? rc : rt_comm_gather_packed(h, c, stripe_rows, 0u, dst_image, stream);
}

int rt_comm_stage_accum(rt_handle h, rt_comm_handle c, uint32_t slot, void* stream)
{
    if (c == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    if (slot > 1u) {
        return fail(c, RT_ERR_BAD_ARGUMENT, "bad argument: slot must be 0 or 1");
    }
    Extent e;
    int rc = image_of(c, h, e);
    if (rc != RT_OK) {
        return rc;
    }
    DeviceScope scope(c->device);
    RC_HIP(c, scope.status());
    const size_t bytes = static_cast<size_t>(e.width) * e.height * sizeof(float4);
    rc = ensure(c, c->staging[slot], c->staging_bytes[slot], bytes);
    if (rc != RT_OK) {
        return rc;
    }
    RC_HIP(c, hipMemcpyAsync(c->staging[slot], e.accum, bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
    return RT_OK;
}

namespace {
int reduce_in_place(rt_comm* c, void* image, size_t texels, hipStream_t s)
{
    RC_NCCL(c, ncclAllReduce(image, image, texels * 4u, ncclFloat, ncclSum, c->comm, s));
    const size_t blocks = (texels + kThreads - 1u) / kThreads;
    hipLaunchKernelGGL(rt_alpha_one_kernel, dim3(static_cast<uint32_t>(blocks)), dim3(kThreads), 0, s, static_cast<float4*>(image), texels);
    RC_HIP(c, hipGetLastError());
    return RT_OK;
}
} // namespace

int rt_comm_reduce_staged(rt_handle h, rt_comm_handle c, uint32_t slot, void** image_out, void* stream)
{
    if (c == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    if (slot > 1u) {
        return fail(c, RT_ERR_BAD_ARGUMENT, "bad argument: slot must be 0 or 1");
    }
    Extent e;
    const int rc = image_of(c, h, e);
    if (rc != RT_OK) {
        return rc;
    }
    const size_t texels = static_cast<size_t>(e.width) * e.height;
    if (c->staging[slot] == nullptr || c->staging_bytes[slot] < texels * sizeof(float4)) {
        return fail(c, RT_ERR_BAD_ARGUMENT, "bad argument: no accumulator was staged in this slot");
    }
    DeviceScope scope(c->device);
    RC_HIP(c, scope.status());
    if (image_out != nullptr) {
        *image_out = c->staging[slot];
    }
    return reduce_in_place(c, c->staging[slot], texels, static_cast<hipStream_t>(stream));
}

int rt_reduce_frames(rt_handle h, rt_comm_handle c, void* dst_image, void* stream)
{
    if (c == nullptr) {
        return RT_ERR_BAD_ARGUMENT;
    }
    if (dst_image == nullptr) {
        return fail(c, RT_ERR_BAD_ARGUMENT, "bad argument: dst_image is NULL");
    }
    Extent e;
    const int rc = image_of(c, h, e);
    if (rc != RT_OK) {
        return rc;
    }
    DeviceScope scope(c->device);
    RC_HIP(c, scope.status());
    const size_t texels = static_cast<size_t>(e.width) * e.height;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dst_image != e.accum) {
        RC_HIP(c, hipMemcpyAsync(dst_image, e.accum, texels * sizeof(float4), hipMemcpyDeviceToDevice, s));
    }
    return reduce_in_place(c, dst_image, texels, s);
}

size_t rt_stripes_packed_bytes(uint32_t width, uint32_t height, uint32_t stripe_rows, uint32_t world)
{
    if (width == 0u || height == 0u || stripe_rows == 0u || world == 0u) {
        return 0;
    }
    return stripe_layout(width, height, stripe_rows, static_cast<int>(world), 0).packed_texels * sizeof(float4);
}

int rt_pack_stripes_to(rt_handle h, uint32_t stripe_rows, uint32_t world, uint32_t rank, void* send, void* stream)
{
    if (send == nullptr || stripe_rows == 0u || world == 0u || rank >= world) {
        return fail(nullptr, RT_ERR_BAD_ARGUMENT, "bad argument: send is NULL, stripe_rows is 0 or rank outside the world");
    }
    Extent e;
    const int rc = image_of(nullptr, h, e);
    if (rc != RT_OK) {
        return rc;
    }
    const StripeLayout l = stripe_layout(e.width, e.height, stripe_rows, static_cast<int>(world), static_cast<int>(rank));
    hipLaunchKernelGGL(rt_pack_stripes_kernel, row_grid(l.width, l.slots * l.stripe_rows), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream), e.accum, static_cast<float4*>(send), l);
    RC_HIP(nullptr, hipGetLastError());
    return RT_OK;
}

int rt_unpack_stripes_from(rt_handle h, uint32_t stripe_rows, uint32_t world, const void* recv, void* dst_image, void* stream)
{
    if (recv == nullptr || dst_image == nullptr || stripe_rows == 0u || world == 0u) {
        return fail(nullptr, RT_ERR_BAD_ARGUMENT, "bad argument: recv/dst_image is NULL, stripe_rows or world is 0");
    }
    Extent e;
    const int rc = image_of(nullptr, h, e);
    if (rc != RT_OK) {
        return rc;
    }
    const StripeLayout l = stripe_layout(e.width, e.height, stripe_rows, static_cast<int>(world), 0);
    hipLaunchKernelGGL(rt_unpack_stripes_kernel, row_grid(l.width, l.height), dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float4*>(recv), static_cast<float4*>(dst_image), l);
    RC_HIP(nullptr, hipGetLastError());
    return RT_OK;
}

const char* rt_comm_last_error(rt_comm_handle c)
{
    return c != nullptr ? c->error.c_str() : g_comm_error.c_str();
}

} // extern "C"

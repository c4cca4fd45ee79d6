// An MI355X backend for the reference's class RayTracer: this file takes the place of src/rt/RayTracer.cpp in
// the reference's build and defines the public methods its header declares (src/rt/RayTracer.h:40-51) on top
// of the C ABI of include/rt_mi355x.h.  The reference's own header is used as it is -- compile with
// -I<reference>/src -I<reference>/external/vulkan/Include -I<this repository>/include and link -lrt_mi355x --
// so App::run (src/core/App.cpp:153-156, :221, :366, :403, :424-448, :539) keeps calling what it calls today.
//
// What becomes of the Vulkan arguments: the swapchain gives the extent (RayTracer.cpp:181-183, :219-221); the
// context, the command buffer and the swapchain image index are not used -- the trace is enqueued on a HIP
// stream instead of being recorded into the caller's command buffer, and the rgba8 image stays in the
// library's own device memory (rt_output_device_ptr).  Presenting it -- rt_read_output into a staging buffer
// and vkCmdCopyBufferToImage, or importing the allocation through VK_KHR_external_memory_fd -- is the host's
// part; rt_mi355x_handle() below hands out what it needs.
//
// Errors: the reference throws std::runtime_error from VK_CHECK (src/util/Check.h:8-19); so does this file,
// with the library's message.  The private members of the class stay unused, its private methods undefined
// (nothing calls them).
#include "rt/RayTracer.h"
#include "vk/Swapchain.h"

#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>

#include "RayTracer_mi355x.h"

namespace {

struct Bound {
    rt_handle handle = nullptr;
    void* stream = nullptr;
};

std::mutex gMutex;
std::unordered_map<const RayTracer*, Bound> gBound; // the class has no member to keep the handle in
int gDevice = 0;

Bound bound_of(const RayTracer* tracer)
{
    std::lock_guard<std::mutex> lock(gMutex);
    auto it = gBound.find(tracer);
    return it != gBound.end() ? it->second : Bound{};
}

rt_handle need(const RayTracer* tracer)
{
    rt_handle h = bound_of(tracer).handle;
    if (h == nullptr) {
        throw std::runtime_error("RayTracer used before create() or after destroy()");
    }
    return h;
}

void check(rt_handle h, int rc, const char* what)
{
    if (rc != RT_OK) {
        throw std::runtime_error(std::string(what) + ": " + rt_last_error(h));
    }
}

} // namespace

rt_handle rt_mi355x_handle(const RayTracer* tracer)
{
    return bound_of(tracer).handle;
}

void rt_mi355x_set_device(int device_id)
{
    std::lock_guard<std::mutex> lock(gMutex);
    gDevice = device_id;
}

void rt_mi355x_set_stream(const RayTracer* tracer, void* hip_stream)
{
    std::lock_guard<std::mutex> lock(gMutex);
    auto it = gBound.find(tracer);
    if (it != gBound.end()) {
        it->second.stream = hip_stream;
    }
}

// src/rt/RayTracer.cpp:179-199: default scene, create-time aim, images of the swapchain's extent
void RayTracer::create(VulkanContext&, Swapchain& swapchain)
{
    const VkExtent2D extent = swapchain.bundle().extent;
    int device = 0;
    {
        std::lock_guard<std::mutex> lock(gMutex);
        if (gBound.count(this) != 0) {
            throw std::runtime_error("RayTracer::create called twice");
        }
        device = gDevice;
    }
    rt_handle h = nullptr;
    if (rt_create(device, extent.width, extent.height, &h) != RT_OK) {
        throw std::runtime_error(std::string("rt_create: ") + rt_last_error(nullptr));
    }
    std::lock_guard<std::mutex> lock(gMutex);
    gBound[this].handle = h;
}

// :201-234: new images of the new extent, accumulation starts over
void RayTracer::resize(VulkanContext&, Swapchain& swapchain)
{
    const VkExtent2D extent = swapchain.bundle().extent;
    rt_handle h = need(this);
    check(h, rt_resize(h, extent.width, extent.height), "rt_resize");
}

// :236-296
void RayTracer::destroy(VulkanContext&)
{
    rt_handle h = nullptr;
    {
        std::lock_guard<std::mutex> lock(gMutex);
        auto it = gBound.find(this);
        if (it == gBound.end()) {
            return;
        }
        h = it->second.handle;
        gBound.erase(it);
    }
    rt_destroy(h);
}

// :298-342: the library applies the same clamps and raises the same reset flag
void RayTracer::setCamera(const glm::vec3& pos, const glm::vec3& dir, float focusDist)
{
    rt_handle h = need(this);
    const float p[3] = { pos.x, pos.y, pos.z }, d[3] = { dir.x, dir.y, dir.z };
    check(h, rt_set_camera(h, p, d, focusDist), "rt_set_camera");
}

void RayTracer::setSamplesPerPixel(uint32_t spp)
{
    rt_handle h = need(this);
    check(h, rt_set_samples_per_pixel(h, spp), "rt_set_samples_per_pixel");
}

void RayTracer::setAperture(float aperture)
{
    rt_handle h = need(this);
    check(h, rt_set_aperture(h, aperture), "rt_set_aperture");
}

void RayTracer::setFocusDistance(float focusDist)
{
    rt_handle h = need(this);
    check(h, rt_set_focus_distance(h, focusDist), "rt_set_focus_distance");
}

void RayTracer::setFov(float vfov)
{
    rt_handle h = need(this);
    check(h, rt_set_fov(h, vfov), "rt_set_fov");
}

void RayTracer::setMaxDepth(uint32_t depth)
{
    rt_handle h = need(this);
    check(h, rt_set_max_depth(h, depth), "rt_set_max_depth");
}

// :559-685: clear iff the reset flag is raised or frameIndex == 0, then one dispatch of the shader
void RayTracer::render(VulkanContext&, Swapchain&, VkCommandBuffer, uint32_t, uint32_t frameIndex)
{
    const Bound b = bound_of(this);
    if (b.handle == nullptr) {
        throw std::runtime_error("RayTracer used before create() or after destroy()");
    }
    check(b.handle, rt_render(b.handle, frameIndex, b.stream), "rt_render");
}

// C++ mirror of the reference's class RayTracer (src/rt/RayTracer.h:34-94) over the C ABI of
// include/rt_mi355x.h.  Same method names, argument meaning and error behaviour: failures throw
// std::runtime_error carrying the library's message, as VK_CHECK does (src/util/Check.h:8-19).
// The Vulkan context / swapchain / command buffer arguments of the reference become a device
// index, an extent and a HIP stream.
#pragma once

#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rt_mi355x.h"

namespace rtamd {

using GPUSphere = rt_gpu_sphere; // src/rt/RayTracer.h:12-17
using GPUParams = rt_gpu_params; // src/rt/RayTracer.h:20-32

struct Extent {
    uint32_t width;
    uint32_t height;
};

class RayTracer {
public:
    RayTracer() = default;
    ~RayTracer() { if (mHandle) { rt_destroy(mHandle); } }
    RayTracer(const RayTracer&) = delete;
    RayTracer& operator=(const RayTracer&) = delete;

    // RayTracer::create(VulkanContext&, Swapchain&): device replaces the context, extent the swapchain's.
    void create(int deviceId, Extent extent)
    {
        if (mHandle) {
            throw std::runtime_error("RayTracer already created");
        }
        int rc = rt_create(deviceId, extent.width, extent.height, &mHandle);
        if (rc != RT_OK) {
            throw std::runtime_error(std::string("rt_create: ") + rt_last_error(nullptr));
        }
    }
    void resize(Extent extent) { check(rt_resize(mHandle, extent.width, extent.height), "rt_resize"); }
    void destroy()
    {
        if (mHandle) {
            rt_destroy(mHandle);
            mHandle = nullptr;
        }
    }
    void setCamera(const std::array<float, 3>& pos, const std::array<float, 3>& dir, float focusDist = -1.0f)
    {
        check(rt_set_camera(mHandle, pos.data(), dir.data(), focusDist), "rt_set_camera");
    }
    void setSamplesPerPixel(uint32_t spp) { check(rt_set_samples_per_pixel(mHandle, spp), "rt_set_samples_per_pixel"); }
    void setAperture(float aperture) { check(rt_set_aperture(mHandle, aperture), "rt_set_aperture"); }
    void setFocusDistance(float focusDist) { check(rt_set_focus_distance(mHandle, focusDist), "rt_set_focus_distance"); }
    void setFov(float vfov) { check(rt_set_fov(mHandle, vfov), "rt_set_fov"); }
    void setMaxDepth(uint32_t depth) { check(rt_set_max_depth(mHandle, depth), "rt_set_max_depth"); }

    // RayTracer::render(ctx, swapchain, commandBuffer, swapImageIndex, frameIndex): the command buffer
    // becomes a HIP stream; the output image is owned by the tracer (outputDevicePtr()).
    void render(uint32_t frameIndex, void* hipStream = nullptr) { check(rt_render(mHandle, frameIndex, hipStream), "rt_render"); }

    // Beyond the reference's public surface.
    void setSpheres(const std::vector<GPUSphere>& spheres)
    {
        check(rt_set_spheres(mHandle, spheres.data(), static_cast<uint32_t>(spheres.size())), "rt_set_spheres");
    }
    // false = test every sphere against every ray (the shader's loop); true (default) = scenes of 24
    // spheres or more go through a bounding-sphere hierarchy with identical results
    void setSceneHierarchy(bool enable) { check(rt_set_scene_hierarchy(mHandle, enable ? 1 : 0), "rt_set_scene_hierarchy"); }
    void renderFrames(uint32_t firstFrame, uint32_t frameCount, void* hipStream = nullptr)
    {
        check(rt_render_frames(mHandle, firstFrame, frameCount, hipStream), "rt_render_frames");
    }
    // Shards for one GPU of several: a row band and/or every frameStride-th frame; every stripeStride-th stripe of
    // stripeRows rows (rank r of G: stripeFirst = r, stripeStride = G).
    void renderShard(uint32_t firstFrame, uint32_t frameCount, uint32_t frameStride, uint32_t rowBegin, uint32_t rowEnd,
                     bool writeOutput = true, void* hipStream = nullptr)
    {
        check(rt_render_shard(mHandle, firstFrame, frameCount, frameStride, rowBegin, rowEnd, writeOutput ? 1 : 0, hipStream),
              "rt_render_shard");
    }
    void renderStripes(uint32_t firstFrame, uint32_t frameCount, uint32_t frameStride, uint32_t stripeRows, uint32_t stripeFirst,
                       uint32_t stripeStride, bool writeOutput = true, void* hipStream = nullptr)
    {
        check(rt_render_stripes(mHandle, firstFrame, frameCount, frameStride, stripeRows, stripeFirst, stripeStride,
                                writeOutput ? 1 : 0, hipStream), "rt_render_stripes");
    }
    // renderFrames for callers that must not allocate device memory (work buffers in the library's static device array).
    void renderFramesStatic(uint32_t firstFrame, uint32_t frameCount, void* hipStream = nullptr)
    {
        check(rt_render_frames_static(mHandle, firstFrame, frameCount, hipStream), "rt_render_frames_static");
    }
    // Consecutive single-frame render() calls overlap on the GPU by default; false keeps every launch on the caller's stream.
    void setFrameOverlap(bool enable) { check(rt_set_frame_overlap(mHandle, enable ? 1 : 0), "rt_set_frame_overlap"); }
    GPUParams params(uint32_t frameIndex) const
    {
        GPUParams p{};
        check(rt_get_params(mHandle, frameIndex, &p), "rt_get_params");
        return p;
    }
    // Blocking copies of the accumulator (float4 per pixel) and of the rgba8 image to the host, row 0 = top.
    std::vector<float> readAccum(Extent extent, void* hipStream = nullptr) const
    {
        std::vector<float> image(static_cast<size_t>(extent.width) * extent.height * 4u);
        check(rt_read_accum(mHandle, image.data(), image.size() * sizeof(float), hipStream), "rt_read_accum");
        return image;
    }
    std::vector<uint8_t> readOutput(Extent extent, void* hipStream = nullptr) const
    {
        std::vector<uint8_t> image(static_cast<size_t>(extent.width) * extent.height * 4u);
        check(rt_read_output(mHandle, image.data(), image.size(), hipStream), "rt_read_output");
        return image;
    }
    void* accumDevicePtr() const { return rt_accum_device_ptr(mHandle); }
    void* outputDevicePtr() const { return rt_output_device_ptr(mHandle); }
    void synchronize(void* hipStream = nullptr) { check(rt_synchronize(mHandle, hipStream), "rt_synchronize"); }
    rt_handle handle() const { return mHandle; }

private:
    void check(int rc, const char* what) const
    {
        if (rc != RT_OK) {
            throw std::runtime_error(std::string(what) + ": " + rt_last_error(mHandle));
        }
    }
    rt_handle mHandle = nullptr;
};

} // namespace rtamd

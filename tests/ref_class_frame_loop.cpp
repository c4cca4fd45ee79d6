// The reference's OWN class declaration (src/rt/RayTracer.h, used where it lies) bound to this library by
// ray-tracing_amd/csrc/RayTracer_mi355x.cpp, driven the way App::run drives it: the same calls, in the same
// order, as tests/cpp_frame_loop.cpp makes through the C++ mirror, and the same checksum lines on stdout.
// Built by oracle/Makefile (target ref) where the reference tree is present; run on the GPU by
// tests/test_gpu_cpp_host.py.  The Vulkan context is never touched by the backend, so none exists here; the
// swapchain is a default-constructed object whose extent is set directly.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include <vulkan/vulkan.h>
#define private public // to set the extent of a swapchain that no Vulkan device ever created
#include "vk/Swapchain.h"
#undef private
#include "rt/RayTracer.h"

#include "../ray-tracing_amd/csrc/RayTracer_mi355x.h"

namespace {

// position-weighted sum of the 32-bit words, modulo 2^64 (the test computes the same with numpy)
uint64_t word_sum(const void* data, size_t bytes)
{
    uint64_t h = 0;
    for (size_t i = 0; i < bytes / 4; ++i) {
        uint32_t w;
        std::memcpy(&w, static_cast<const unsigned char*>(data) + 4 * i, 4);
        h += static_cast<uint64_t>(w) * (2 * i + 1);
    }
    return h;
}

int report(const char* stage, const RayTracer& tracer, VkExtent2D extent)
{
    rt_handle h = rt_mi355x_handle(&tracer);
    std::vector<float> accum(static_cast<size_t>(extent.width) * extent.height * 4u);
    std::vector<uint8_t> output(static_cast<size_t>(extent.width) * extent.height * 4u);
    if (rt_read_accum(h, accum.data(), accum.size() * sizeof(float), nullptr) != RT_OK ||
        rt_read_output(h, output.data(), output.size(), nullptr) != RT_OK) {
        std::printf("error: %s\n", rt_last_error(h));
        return 1;
    }
    std::printf("%s %016llx %016llx\n", stage, (unsigned long long)word_sum(accum.data(), accum.size() * sizeof(float)),
                (unsigned long long)word_sum(output.data(), output.size()));
    return 0;
}

} // namespace

int main()
{
    alignas(16) static unsigned char no_context[16];
    VulkanContext& context = *reinterpret_cast<VulkanContext*>(no_context); // an incomplete type here; never used
    try {
        Swapchain swapchain;
        swapchain.mSwapchainBundle.extent = { 320, 180 };
        RayTracer tracer;
        rt_mi355x_set_device(0);
        tracer.create(context, swapchain);
        rt_mi355x_set_stream(&tracer, nullptr); // the null stream, which is also the default
        tracer.setSamplesPerPixel(4);
        tracer.setAperture(0.05f);
        uint32_t frame = 0;
        for (int i = 0; i < 6; ++i) {
            tracer.render(context, swapchain, VK_NULL_HANDLE, 0, frame++);
        }
        if (report("first_view", tracer, swapchain.bundle().extent)) { return 1; }

        tracer.setCamera({ 10.0f, 3.0f, 6.0f }, { -0.8f, -0.2f, -0.5f });
        frame = 0;
        for (int i = 0; i < 3; ++i) {
            tracer.render(context, swapchain, VK_NULL_HANDLE, 0, frame++);
        }
        if (report("camera_moved", tracer, swapchain.bundle().extent)) { return 1; }

        tracer.setMaxDepth(5);
        tracer.setFov(35.0f);
        tracer.setFocusDistance(9.0f);
        frame = 0;
        for (int i = 0; i < 2; ++i) {
            tracer.render(context, swapchain, VK_NULL_HANDLE, 0, frame++);
        }
        if (report("settings_changed", tracer, swapchain.bundle().extent)) { return 1; }

        swapchain.mSwapchainBundle.extent = { 200, 120 };
        tracer.resize(context, swapchain);
        frame = 0;
        tracer.render(context, swapchain, VK_NULL_HANDLE, 0, frame++);
        // the batched extension, reached through the handle, continues the same accumulation
        if (rt_render_frames(rt_mi355x_handle(&tracer), frame, 4, nullptr) != RT_OK) {
            std::printf("error: %s\n", rt_last_error(rt_mi355x_handle(&tracer)));
            return 1;
        }
        if (report("resized", tracer, swapchain.bundle().extent)) { return 1; }
        tracer.destroy(context);
        if (rt_mi355x_handle(&tracer) != nullptr) {
            std::printf("error: the handle outlived destroy()\n");
            return 1;
        }
    } catch (const std::exception& e) {
        std::printf("error: %s\n", e.what());
        return 1;
    }
    return 0;
}

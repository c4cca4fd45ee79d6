// Built by tests/Makefile, run by tests/test_gpu_cpp_host.py on the GPU: a C++ caller that drives the
// mirror of the reference class the way App::run drives RayTracer (src/core/App.cpp:153-156 create and
// first settings, :403 one render per frame, :366 setCamera and :173/:527 frame counter back to 0 when
// the camera moves, :424-448 settings, :221 resize), and prints checksums of the accumulator and of
// the rgba8 image after each stage.  The test replays the same calls through the Python mirror, whose
// images are checked against the CPU oracle, and compares the checksums.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../ray-tracing_amd/csrc/RayTracer.hpp"

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

void report(const char* stage, const rtamd::RayTracer& tracer, rtamd::Extent extent)
{
    const std::vector<float> accum = tracer.readAccum(extent);
    const std::vector<uint8_t> output = tracer.readOutput(extent);
    std::printf("%s %016llx %016llx\n", stage, (unsigned long long)word_sum(accum.data(), accum.size() * sizeof(float)),
                (unsigned long long)word_sum(output.data(), output.size()));
}

} // namespace

int main()
{
    try {
        rtamd::Extent extent{ 320, 180 };
        rtamd::RayTracer tracer;
        tracer.create(0, extent);
        tracer.setSamplesPerPixel(4);
        tracer.setAperture(0.05f);
        uint32_t frame = 0;
        for (int i = 0; i < 6; ++i) {
            tracer.render(frame++);
        }
        report("first_view", tracer, extent);

        tracer.setCamera({ 10.0f, 3.0f, 6.0f }, { -0.8f, -0.2f, -0.5f });
        frame = 0;
        for (int i = 0; i < 3; ++i) {
            tracer.render(frame++);
        }
        report("camera_moved", tracer, extent);

        tracer.setMaxDepth(5);
        tracer.setFov(35.0f);
        tracer.setFocusDistance(9.0f);
        frame = 0;
        for (int i = 0; i < 2; ++i) {
            tracer.render(frame++);
        }
        report("settings_changed", tracer, extent);

        extent = { 200, 120 };
        tracer.resize(extent);
        frame = 0;
        tracer.render(frame++);
        tracer.renderFrames(frame, 4); // the batched extension continues the same accumulation
        report("resized", tracer, extent);
        tracer.destroy();
    } catch (const std::exception& e) {
        std::printf("error: %s\n", e.what());
        return 1;
    }
    return 0;
}

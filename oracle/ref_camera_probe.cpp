// Golden-vector generator for the caller-side camera arithmetic -- TEST INFRASTRUCTURE ONLY.
//
// The frame loop of the reference (src/core/App.cpp:180-183, :299-362) turns mouse deltas into yaw and
// pitch, those into a direction, and key presses into steps of the camera position, all with the
// reference's vendored GLM 1.0.1 and the C library's cosf/sinf/atan2/asin.  App.cpp itself cannot be
// compiled here (it needs GLFW, ImGui and a Vulkan loader), so this probe evaluates the same expressions,
// in the same order, with the reference's own copy of GLM (included where it lies) on a scripted event
// sequence and prints every intermediate state as hex words.  tests/golden/make_golden.py turns the
// output into tests/golden/caller_camera.json, which pins ray-tracing_amd/caller.py bit for bit.
//
// Build without contraction: g++ -O2 -ffp-contract=off -DGLM_FORCE_PURE.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include <glm/glm.hpp>

static uint32_t bits_of(float f)
{
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

struct Event {
    double dx, dy;     // cursor deltas (deltaX, deltaY of App.cpp:299-300)
    const char* keys;  // subset of "WSADUP" (U = space, P = left shift), in any order
    double dt;         // deltaTime
};

int main()
{
    // App.cpp:180-183
    glm::vec3 camPos{ 13.0f, 2.0f, 3.0f };
    glm::vec3 camDir = glm::normalize(glm::vec3(0.0f, 1.0f, 0.0f) - camPos);
    float yaw = glm::degrees(std::atan2(camDir.z, camDir.x));
    float pitch = glm::degrees(std::asin(camDir.y));
    std::printf("init %08x %08x %08x %08x %08x %08x %08x %08x\n", bits_of(camPos.x), bits_of(camPos.y), bits_of(camPos.z),
                bits_of(camDir.x), bits_of(camDir.y), bits_of(camDir.z), bits_of(yaw), bits_of(pitch));

    const Event script[] = {
        { 25.0, 5.0, "", 1.0 / 60.0 },      { 0.0, 0.0, "WD", 0.1 },          { -130.5, 12.25, "W", 0.016 },
        { 3.0, -400.0, "SA", 0.033 },       { 1.0, 2000.0, "U", 0.0167 },     { 0.0, -0.5, "P", 0.25 },
        { 719.0, -33.0, "WSADUP", 0.0069 }, { -0.125, 0.0, "A", 1.0 / 144.0 }, { 10000.0, 1.0, "D", 0.5 },
        { 7.0, 7.0, "", 0.02 },             { 0.0, 0.0, "S", 0.011 },         { -45.0, -45.0, "WU", 0.0042 },
    };
    for (const Event& e : script) {
        // App.cpp:299-322
        const float sensitivity = 0.1f;
        if (e.dx != 0.0 || e.dy != 0.0) {
            yaw += static_cast<float>(e.dx) * sensitivity;
            pitch += static_cast<float>(e.dy) * sensitivity;
            pitch = std::clamp(pitch, -89.0f, 89.0f);
            float yawRad = glm::radians(yaw);
            float pitchRad = glm::radians(pitch);
            glm::vec3 newDir{ cosf(pitchRad) * cosf(yawRad), sinf(pitchRad), cosf(pitchRad) * sinf(yawRad) };
            camDir = glm::normalize(newDir);
        }
        // App.cpp:326-362
        glm::vec3 right = glm::normalize(glm::cross(camDir, glm::vec3(0.0f, 1.0f, 0.0f)));
        glm::vec3 up = glm::vec3(0.0f, 1.0f, 0.0f);
        const float moveSpeed = 5.0f;
        const double deltaTime = e.dt;
        auto pressed = [&](char k) { return std::strchr(e.keys, k) != nullptr; };
        if (pressed('W')) { camPos += camDir * moveSpeed * static_cast<float>(deltaTime); }
        if (pressed('S')) { camPos -= camDir * moveSpeed * static_cast<float>(deltaTime); }
        if (pressed('A')) { camPos -= right * moveSpeed * static_cast<float>(deltaTime); }
        if (pressed('D')) { camPos += right * moveSpeed * static_cast<float>(deltaTime); }
        if (pressed('U')) { camPos += up * moveSpeed * static_cast<float>(deltaTime); }
        if (pressed('P')) { camPos -= up * moveSpeed * static_cast<float>(deltaTime); }
        std::printf("step %.17g %.17g %s %.17g : %08x %08x %08x %08x %08x %08x %08x %08x\n", e.dx, e.dy,
                    e.keys[0] ? e.keys : "-", e.dt, bits_of(camPos.x), bits_of(camPos.y), bits_of(camPos.z),
                    bits_of(camDir.x), bits_of(camDir.y), bits_of(camDir.z), bits_of(yaw), bits_of(pitch));
    }
    return 0;
}

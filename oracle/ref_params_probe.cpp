// Golden-vector generator for the HOST half of the path -- TEST INFRASTRUCTURE ONLY.
//
// Links against the reference's own src/rt/RayTracer.cpp (compiled where it
// lies under /root/reference by oracle/Makefile; nothing is copied) and drives
// its private buildScene()/makeCameraParams() plus the public setters, then
// prints the resulting GPUSphere / GPUParams bytes as hex words.  The output is
// turned into tests/golden/*.json by tests/golden/make_golden.py.
//
// Usage: ref_params_probe <width> <height> <aperture> <vfov> <focus|-1> <px> <py> <pz> <dx> <dy> <dz> | "default" <width> <height> <aperture>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <vulkan/vulkan.h>
#include <glm/glm.hpp>
#include "vma/vk_mem_alloc.h"

#define private public
#include "rt/RayTracer.h"
#undef private

static void dump(const char* tag, const void* data, size_t bytes)
{
    const uint32_t* words = static_cast<const uint32_t*>(data);
    std::printf("%s", tag);
    for (size_t i = 0; i < bytes / 4; ++i)
    {
        std::printf(" %08x", words[i]);
    }
    std::printf("\n");
}

int main(int argc, char** argv)
{
    if (argc < 5)
    {
        std::fprintf(stderr, "usage: see source\n");
        return 2;
    }

    RayTracer tracer;
    tracer.buildScene();
    {
        // create()'s aim block, src/rt/RayTracer.cpp:189-193, through public state.
        glm::vec3 lookAt{ 0.0f, 1.0f, 0.0f };
        tracer.mCamDir = glm::normalize(lookAt - tracer.mCamPos);
        tracer.mFocusDistance = glm::length(lookAt - tracer.mCamPos);
    }

    uint32_t width = 0, height = 0;
    if (std::strcmp(argv[1], "default") == 0)
    {
        width = static_cast<uint32_t>(std::atoi(argv[2]));
        height = static_cast<uint32_t>(std::atoi(argv[3]));
        tracer.setSamplesPerPixel(4);
        tracer.setAperture(static_cast<float>(std::atof(argv[4])));
    }
    else
    {
        if (argc < 12)
        {
            return 2;
        }
        width = static_cast<uint32_t>(std::atoi(argv[1]));
        height = static_cast<uint32_t>(std::atoi(argv[2]));
        tracer.setAperture(static_cast<float>(std::atof(argv[3])));
        tracer.setFov(static_cast<float>(std::atof(argv[4])));
        float focus = static_cast<float>(std::atof(argv[5]));
        glm::vec3 pos{ std::atof(argv[6]), std::atof(argv[7]), std::atof(argv[8]) };
        glm::vec3 dir{ std::atof(argv[9]), std::atof(argv[10]), std::atof(argv[11]) };
        tracer.setCamera(pos, dir, focus);
    }

    VkExtent2D extent{ width, height };
    GPUParams params = tracer.makeCameraParams(extent);

    std::printf("sizeof_sphere %zu sizeof_params %zu nspheres %zu\n", sizeof(GPUSphere), sizeof(GPUParams), tracer.mSpheres.size());
    dump("spheres", tracer.mSpheres.data(), tracer.mSpheres.size() * sizeof(GPUSphere));
    dump("params", &params, sizeof(GPUParams));
    dump("focus", &tracer.mFocusDistance, 4);
    dump("camdir", &tracer.mCamDir, 12);
    dump("fov", &tracer.mVerticalFov, 4);
    dump("aperture", &tracer.mAperture, 4);
    return 0;
}

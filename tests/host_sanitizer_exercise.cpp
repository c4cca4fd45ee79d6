// Built by tests/Makefile with AddressSanitizer and UBSan on the HOST code (hipcc -Xarch_host -fsanitize=...), run by
// tests/test_host_sanitizers.py: the C ABI on handles without a device -- scenes of many sizes and kinds through both
// hierarchy builders and their getters, parameter blocks, clamped setters, error paths.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../include/rt_mi355x.h"

static uint32_t lcg = 12345u;
static float rnd() { lcg = lcg * 1664525u + 1013904223u; return (lcg >> 8) * (1.0f / 16777216.0f); }

int main()
{
    rt_handle h = nullptr;
    if (rt_create(RT_DEVICE_NONE, 640, 360, &h) != RT_OK) { std::printf("create failed: %s\n", rt_last_error(nullptr)); return 1; }
    rt_gpu_params p;
    for (uint32_t f = 0; f < 4; ++f) { rt_get_params(h, f, &p); }
    const float pos[3] = { 3, 2, 1 }, dir[3] = { -1, -0.2f, 0.3f };
    rt_set_camera(h, pos, dir, -1.0f);
    rt_set_fov(h, 500.0f); rt_set_fov(h, -1.0f); rt_set_aperture(h, -3.0f); rt_set_focus_distance(h, 0.0f);
    rt_set_samples_per_pixel(h, 0); rt_set_max_depth(h, 0);
    const uint32_t sizes[] = { 1, 5, 16, 23, 24, 25, 63, 64, 65, 255, 256, 257, 1000, 4096, 4097, 6000, 7000, 20000, 65536, 300000 };
    for (uint32_t n : sizes) {
        std::vector<rt_gpu_sphere> s(n);
        for (uint32_t i = 0; i < n; ++i) {
            s[i].center_radius[0] = (rnd() - 0.5f) * 200.0f; s[i].center_radius[1] = rnd() * 3.0f; s[i].center_radius[2] = (rnd() - 0.5f) * 200.0f;
            s[i].center_radius[3] = (i % 97 == 0) ? 50.0f * rnd() + 1.0f : 0.05f + 0.3f * rnd();
            if (i % 501 == 3) { s[i].center_radius[3] = -s[i].center_radius[3]; }       // hollow spheres
            if (n == 257 && i == 100) { s[i].center_radius[0] = NAN; }                   // a scene that gets no hierarchy
            s[i].albedo[0] = rnd(); s[i].albedo[1] = rnd(); s[i].albedo[2] = rnd(); s[i].albedo[3] = 0.0f;
            s[i].misc[0] = (float)(i % 3); s[i].misc[1] = rnd(); s[i].misc[2] = 1.5f; s[i].misc[3] = (float)(i % 2);
        }
        // coincident and concentric spheres
        if (n >= 64) { s[10] = s[11]; s[12] = s[13]; s[12].center_radius[3] *= 0.5f; }
        int rc = rt_set_spheres(h, s.data(), n);
        if (rc != RT_OK) { std::printf("set_spheres(%u) failed: %s\n", n, rt_last_error(h)); return 1; }
        uint32_t count = 0;
        std::vector<rt_gpu_sphere> back(n);
        rt_get_spheres(h, back.data(), n, &count);
        for (int mode = 0; mode <= 2; ++mode) {
            rt_set_scene_hierarchy(h, mode);
            uint32_t nodes = 0;
            rt_get_scene_hierarchy(h, nullptr, nullptr, 0, &nodes);
            std::vector<float> geom((size_t)nodes * 4 + 4);
            std::vector<uint32_t> word(nodes + 1), link(nodes + 1), sph(nodes + 1);
            rt_get_scene_hierarchy(h, geom.data(), word.data(), nodes, &nodes);
            rt_get_scene_hierarchy_links(h, geom.data(), link.data(), sph.data(), nodes, &nodes);
        }
        {
            uint32_t groups = 0, slots = 0, levels = 0, roots = 0, first_leaf = 0, prologue = 0; float bound[5];
            rt_get_scene_quadtree(h, nullptr, nullptr, nullptr, 0, 0, 0, &groups, &slots, &levels, &roots, &first_leaf, &prologue, bound);
            std::vector<float> box((size_t)first_leaf * 24 + 24), geom((size_t)slots * 4 + 16);
            std::vector<uint32_t> slot_sphere(slots + 4);
            rt_get_scene_quadtree(h, box.data(), geom.data(), slot_sphere.data(), first_leaf, groups, slots, &groups, &slots, &levels, &roots, &first_leaf, &prologue, bound);
        }
        std::printf("n=%u ok\n", n);
    }
    rt_set_spheres(h, nullptr, 0);
    rt_resize(h, 0, 0); rt_resize(h, 70000, 2); rt_resize(h, 1920, 1080);
    rt_render(h, 0, nullptr);            // no device: an error, not a crash
    rt_destroy(h);
    rt_destroy(nullptr);
    std::printf("done\n");
    return 0;
}

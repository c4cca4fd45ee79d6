// Golden-vector generator for the GLSL built-ins the shader relies on -- TEST INFRASTRUCTURE ONLY.
//
// The reference vendors GLM 1.0.1 (external/vulkan/Include/glm), which implements the GLSL
// built-in functions with the formulas of the GLSL specification.  This probe evaluates, with the
// reference's own copy of GLM (included where it lies, nothing is copied), the built-ins that
// shaders/raytrace.comp.glsl calls -- dot, normalize, inversesqrt, reflect, mix, mod, floor, min,
// max, length, cross -- on a fixed pseudo-random input set and prints inputs and results as hex
// words.  tests/golden/make_golden.py turns the output into tests/golden/glm_builtins.json, which
// pins the oracle's own helpers (rto_builtin) bit for bit: operand order of dot, normalize as
// v * inversesqrt(dot(v, v)), inversesqrt as 1 / sqrt, reflect as I - N * dot(N, I) * 2, mix as
// x * (1 - a) + y * a, mod as x - y * floor(x / y).
//
// Build without contraction and without GLM's SIMD paths so that every operation is one binary32
// rounding: g++ -O2 -ffp-contract=off -DGLM_FORCE_PURE.
#include <cstdint>
#include <cstdio>
#include <cstring>

#include <glm/glm.hpp>

static uint32_t state = 0x2545F491u;
static uint32_t next_u32()
{
    state = 1664525u * state + 1013904223u;
    return state;
}
// values over many binades, both signs
static float next_float(int min_exp, int max_exp)
{
    const uint32_t r = next_u32();
    const int e = min_exp + (int)((r >> 8) % (uint32_t)(max_exp - min_exp + 1));
    const uint32_t mant = next_u32() & 0x007FFFFFu;
    const uint32_t sign = (next_u32() >> 31) << 31;
    const uint32_t bits = sign | ((uint32_t)(e + 127) << 23) | mant;
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}
static uint32_t bits_of(float f)
{
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}
static void emit(const char* op, const float* in, int n_in, const float* out, int n_out)
{
    std::printf("%s", op);
    for (int i = 0; i < n_in; ++i) {
        std::printf(" %08x", bits_of(in[i]));
    }
    std::printf(" ->");
    for (int i = 0; i < n_out; ++i) {
        std::printf(" %08x", bits_of(out[i]));
    }
    std::printf("\n");
}

int main()
{
    for (int k = 0; k < 200; ++k) {
        float in[7], out[3];
        for (int i = 0; i < 7; ++i) {
            in[i] = next_float(-6, 6);
        }
        const glm::vec3 a(in[0], in[1], in[2]), b(in[3], in[4], in[5]);
        out[0] = glm::dot(a, b);
        emit("dot", in, 6, out, 1);
        glm::vec3 v = glm::normalize(a);
        out[0] = v.x; out[1] = v.y; out[2] = v.z;
        emit("normalize", in, 3, out, 3);
        v = glm::reflect(a, b);
        out[0] = v.x; out[1] = v.y; out[2] = v.z;
        emit("reflect", in, 6, out, 3);
        v = glm::cross(a, b);
        out[0] = v.x; out[1] = v.y; out[2] = v.z;
        emit("cross", in, 6, out, 3);
        out[0] = glm::length(a);
        emit("length", in, 3, out, 1);
        // mix(vec3, vec3, float) with a in [0, 1] as the sky gradient uses it, and outside too
        float t = (k % 2 == 0) ? (float)(next_u32() & 0xFFFFFF) / 16777216.0f : in[6];
        float mix_in[7] = { in[0], in[1], in[2], in[3], in[4], in[5], t };
        v = glm::mix(a, b, t);
        out[0] = v.x; out[1] = v.y; out[2] = v.z;
        emit("mix", mix_in, 7, out, 3);
        const float x = next_float(-3, 12), y = (k % 3 == 0) ? 2.0f : next_float(-2, 4);
        float mod_in[2] = { x, y };
        out[0] = glm::mod(x, y);
        emit("mod", mod_in, 2, out, 1);
        out[0] = glm::floor(x);
        emit("floor", mod_in, 1, out, 1);
        const float p = glm::abs(next_float(-20, 20));
        out[0] = glm::inversesqrt(p);
        emit("inversesqrt", &p, 1, out, 1);
        out[0] = glm::min(x, y);
        emit("min", mod_in, 2, out, 1);
        out[0] = glm::max(x, y);
        emit("max", mod_in, 2, out, 1);
        // pow is implementation-defined in GLSL; recorded for a tolerance check only
        const float base = glm::abs(next_float(-8, 0)), expo = (k % 2 == 0) ? 5.0f : 1.0f / 2.2f;
        float pow_in[2] = { base, expo };
        out[0] = glm::pow(base, expo);
        emit("pow", pow_in, 2, out, 1);
    }
    return 0;
}

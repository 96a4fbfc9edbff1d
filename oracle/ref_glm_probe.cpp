// ref_glm_probe.cpp — golden-vector generator (test infrastructure).
//
// Compiles the REFERENCE's own vendored math — GLM 0.9.9.7 under
// /root/reference/Libraries/include/glm and main/src/utilities/glm_utils.h —
// from where it lies (oracle/Makefile `ref` target, output in oracle/_ref/),
// evaluates every GLM / glm_utils function the hot path uses on seeded inputs
// that include IEEE special values, and writes inputs + outputs as raw
// little-endian fp32 arrays under tests/golden/.  These are the only files of
// the reference on this path that build in this image without stand-in
// headers (everything else includes cuda_runtime.h / curand_kernel.h).
//
// Usage: oracle/_ref/glm_probe <out_dir>     (see oracle/gen_golden.py)
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include <glm/glm.hpp>
#include <glm/gtx/component_wise.hpp>
#include "utilities/glm_utils.h"

static uint64_t g_state = 0x1984ull;
static uint32_t next_u32() {  // splitmix64
    uint64_t z = (g_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)((z ^ (z >> 31)) >> 32);
}
static float uni() { return (float)(next_u32() >> 8) * (1.0f / 16777216.0f); }
static float special(uint32_t k) {
    const float inf = std::numeric_limits<float>::infinity();
    const float nan = std::numeric_limits<float>::quiet_NaN();
    const float tab[] = {0.0f, -0.0f, 1.0f, -1.0f, inf, -inf, nan, 1e-9f, -1e-9f, 1.0000001e-9f, 1e-38f, 1e-42f,
                         -1e-42f, 3.402823466e+38F, -3.402823466e+38F, 1e20f, -1e20f, 0.5f, 2.0f, 1e9f, -1e9f};
    return tab[k % (sizeof(tab) / sizeof(tab[0]))];
}
// mostly "scene scale" values, sometimes wide-range, sometimes special
static float val() {
    uint32_t r = next_u32() % 16;
    if (r == 0) return special(next_u32());
    if (r == 1) return std::ldexp(uni() * 2.0f - 1.0f, (int)(next_u32() % 80) - 40);
    if (r == 2) return (float)((int)(next_u32() % 7) - 3);
    return (uni() * 2.0f - 1.0f) * 20.0f;
}
static glm::vec3 v3() { return glm::vec3(val(), val(), val()); }
static glm::vec3 unit3() {
    glm::vec3 v(uni() * 2 - 1, uni() * 2 - 1, uni() * 2 - 1);
    if (glm::length2(v) < 1e-6f) v = glm::vec3(1, 0, 0);
    return glm::normalize(v);
}

struct Sink {
    std::string dir;
    std::vector<float> in, out;
    void i(float f) { in.push_back(f); }
    void i(const glm::vec3& v) { in.push_back(v.x); in.push_back(v.y); in.push_back(v.z); }
    void o(float f) { out.push_back(f); }
    void o(const glm::vec3& v) { out.push_back(v.x); out.push_back(v.y); out.push_back(v.z); }
    void flush(const char* name) {
        auto wr = [&](const std::string& p, const std::vector<float>& d) {
            FILE* f = std::fopen(p.c_str(), "wb");
            if (!f) { std::perror(p.c_str()); std::exit(1); }
            std::fwrite(d.data(), sizeof(float), d.size(), f);
            std::fclose(f);
        };
        wr(dir + "/glm_" + name + "_in.f32", in);
        wr(dir + "/glm_" + name + "_out.f32", out);
        std::printf("%-14s in %zu floats, out %zu floats\n", name, in.size(), out.size());
        in.clear(); out.clear();
    }
};

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s <out_dir>\n", argv[0]); return 2; }
    Sink s; s.dir = argv[1];
    const int N = 1024;

    for (int k = 0; k < N; k++) { auto a = v3(), b = v3(); s.i(a); s.i(b); s.o(glm::dot(a, b)); }
    s.flush("dot");
    for (int k = 0; k < N; k++) { auto a = v3(), b = v3(); s.i(a); s.i(b); s.o(glm::cross(a, b)); }
    s.flush("cross");
    for (int k = 0; k < N; k++) { auto a = v3(); s.i(a); s.o(glm::normalize(a)); }
    s.flush("normalize");
    for (int k = 0; k < N; k++) {
        auto i = v3(); auto n = (k & 1) ? unit3() : v3();
        s.i(i); s.i(n); s.o(glm::reflect(i, n));
    }
    s.flush("reflect");
    for (int k = 0; k < N; k++) {
        auto i = (k & 3) ? unit3() : v3(); auto n = (k & 3) ? unit3() : v3();
        float eta = (k % 3 == 0) ? 1.5f : ((k % 3 == 1) ? 1.0f / 1.5f : val());
        s.i(i); s.i(n); s.i(eta); s.o(glm::refract(i, n, eta));
    }
    s.flush("refract");
    for (int k = 0; k < N; k++) {
        auto a = v3(), b = v3(); float t = (k & 1) ? uni() : val();
        s.i(a); s.i(b); s.i(t); s.o(glm::mix(a, b, t));
    }
    s.flush("mix3");
    for (int k = 0; k < N; k++) {
        float a = val(), b = val(), t = (k & 1) ? uni() : val();
        s.i(a); s.i(b); s.i(t); s.o(glm::mix(a, b, t));
    }
    s.flush("mix1");
    for (int k = 0; k < N; k++) { auto a = v3(), b = v3(); s.i(a); s.i(b); s.o(glm::min(a, b)); }
    s.flush("min3");
    for (int k = 0; k < N; k++) { auto a = v3(), b = v3(); s.i(a); s.i(b); s.o(glm::max(a, b)); }
    s.flush("max3");
    for (int k = 0; k < N; k++) { auto a = v3(); s.i(a); s.o(glm::compMax(a)); }
    s.flush("compmax");
    for (int k = 0; k < N; k++) { auto a = v3(); s.i(a); s.o(glm::compMin(a)); }
    s.flush("compmin");
    // Renderer.cu:209-211: clamp then sqrt
    for (int k = 0; k < N; k++) {
        auto a = (k & 1) ? glm::vec3(uni() * 1.5f - 0.25f, uni() * 1.5f - 0.25f, uni() * 1.5f - 0.25f) : v3();
        s.i(a); s.o(glm::sqrt(glm::clamp(a, 0.0f, 1.0f)));
    }
    s.flush("clamp01_sqrt");
    for (int k = 0; k < N; k++) {
        glm::vec3 a = (k & 1) ? glm::vec3(special(next_u32()), special(next_u32()), special(next_u32()))
                              : glm::vec3(std::ldexp(uni() - 0.5f, -28), std::ldexp(uni() - 0.5f, -28), std::ldexp(uni() - 0.5f, -28));
        s.i(a); s.o(glm::near_zero(a) ? 1.0f : 0.0f);
    }
    s.flush("near_zero");
    for (int k = 0; k < N; k++) { auto a = v3(); s.i(a); s.o(glm::length2(a)); }
    s.flush("length2");
    for (int k = 0; k < N; k++) {
        auto a = v3(), b = v3(); float t = (k & 1) ? uni() : val();
        s.i(a); s.i(b); s.i(t); s.o(glm::linear_interpolate(a, b, t));
    }
    s.flush("lerp");
    for (int k = 0; k < N; k++) { float d = (k & 1) ? uni() * 180.0f : val(); s.i(d); s.o(glm::radians(d)); }
    s.flush("radians");
    return 0;
}

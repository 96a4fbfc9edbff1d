// rt_math.hpp — fp32 vector vocabulary shared by the host scene code and the HIP kernels.
//
// The reference does all its math through GLM (glm::vec3, vendored 0.9.9.7); the hot path needs a
// dozen functions of it.  They are re-implemented here as a tiny POD `f3` with GLM's exact
// evaluation order and NaN semantics, because CPU/GPU parity on this path is a bit-exactness
// problem (one flipped hit/miss or rejection-loop branch moves a pixel by 1/spp).  Everything is
// compiled with -ffp-contract=off (no FMA contraction) and correctly rounded / and sqrtf.
// Citations: glm/ = /root/reference/Libraries/include/glm, utils = main/src/utilities/glm_utils.h.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#define RT_HD __host__ __device__ __forceinline__

#define RT_MISS_DIST 3.402823466e+38F  // _MISS_DIST, rt_engine/ray_data.cuh:17

struct f3 {
    float x, y, z;
};

RT_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_HD f3 mk3(float s) { return mk3(s, s, s); }
RT_HD f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
RT_HD void st3(float* p, f3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
RT_HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_HD f3 operator/(f3 a, f3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
RT_HD f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
RT_HD f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
RT_HD f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }

// glm/detail/func_common.inl:17-30 — NOT fminf/fmaxf: a NaN in x is returned, a NaN in y is dropped.
RT_HD float glm_min(float x, float y) { return (y < x) ? y : x; }
RT_HD float glm_max(float x, float y) { return (x < y) ? y : x; }
RT_HD f3 glm_min(f3 a, f3 b) { return mk3(glm_min(a.x, b.x), glm_min(a.y, b.y), glm_min(a.z, b.z)); }
RT_HD f3 glm_max(f3 a, f3 b) { return mk3(glm_max(a.x, b.x), glm_max(a.y, b.y), glm_max(a.z, b.z)); }
// glm/gtx/component_wise.inl:111-126
RT_HD float comp_min(f3 a) { float r = a.x; r = glm_min(r, a.y); r = glm_min(r, a.z); return r; }
RT_HD float comp_max(f3 a) { float r = a.x; r = glm_max(r, a.y); r = glm_max(r, a.z); return r; }
// glm/detail/func_geometric.inl:48-56
RT_HD float dot(f3 a, f3 b) { f3 t = a * b; return t.x + t.y + t.z; }
// glm/detail/func_geometric.inl:70-81
RT_HD f3 cross(f3 x, f3 y) { return mk3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }
// glm/detail/func_geometric.inl:84-93, func_exponential.inl:134-139
RT_HD f3 normalize(f3 a) { float inv = 1.0f / sqrtf(dot(a, a)); return a * inv; }
// glm/detail/func_geometric.inl:104-110
RT_HD f3 reflect(f3 i, f3 n) { return i - n * dot(n, i) * 2.0f; }
// glm/detail/func_geometric.inl:113-123
RT_HD f3 refract(f3 i, f3 n, float eta) {
    float dv = dot(n, i);
    float k = 1.0f - eta * eta * (1.0f - dv * dv);
    if (k >= 0.0f) return i * eta - n * (eta * dv + sqrtf(k));
    return mk3(0.0f);
}
// glm/detail/func_common.inl:104-112,124-132
RT_HD f3 mix(f3 x, f3 y, float a) { return x * (1.0f - a) + y * a; }
RT_HD float mix(float x, float y, float a) { return x * (1.0f - a) + y * a; }
// utils:15-25 (epsilon 1e-9f)
RT_HD bool near_zero(f3 a) { return !(fabsf(a.x) > 1e-9f) && !(fabsf(a.y) > 1e-9f) && !(fabsf(a.z) > 1e-9f); }
// utils:27-35
RT_HD float length2(f3 a) { float s = 0.0f; s += a.x * a.x; s += a.y * a.y; s += a.z * a.z; return s; }
RT_HD float length2(float x, float y) { float s = 0.0f; s += x * x; s += y * y; return s; }
// utils:64-67
RT_HD f3 linear_interpolate(f3 a, f3 b, float f) { return a + (b - a) * f; }
// glm/detail/func_trigonometric.inl:9-14
RT_HD float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }
// Renderer.cu:209-211: glm::clamp(x,0,1) = min(max(x,0),1), then glm::sqrt
RT_HD f3 clamp01_sqrt(f3 a) {
    f3 c = glm_min(glm_max(a, mk3(0.0f)), mk3(1.0f));
    return mk3(sqrtf(c.x), sqrtf(c.y), sqrtf(c.z));
}

// ---------------------------------------------------------------------------------------------
// Deterministic fp32 elementary functions for the extension materials (constant media, Perlin marble, sphere uv).
// The reference has none of these features; what matters here is that the CPU oracle and the GPU compute the SAME
// bits, which libm / ocml do not promise.  So: plain fp32 +, -, *, /, sqrt, floor and integer bit operations only,
// in one fixed order, no fused multiply-add (both sides build with -ffp-contract=off).  Accuracy is a few ulp.
// ---------------------------------------------------------------------------------------------
RT_HD float rt_logf(float x) {  // x > 0, finite, normal
    uint32_t ix = __builtin_bit_cast(uint32_t, x);
    int e = (int)(ix >> 23) - 127;
    float m = __builtin_bit_cast(float, (ix & 0x007fffffu) | 0x3f800000u);  // [1, 2)
    if (m > 0x1.6a09e6p+0f) { m = m * 0.5f; e += 1; }     // [sqrt(1/2), sqrt(2))
    float f = m - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    float p = z * (0x1.555556p-1f + z * (0x1.99999ap-2f + z * (0x1.24924ap-2f + z * 0x1.c71c72p-3f)));  // 2/3, 2/5, 2/7, 2/9
    float lm = s * (2.0f + p);                             // log(m) = 2 atanh(s)
    float fe = (float)e;
    return fe * 0x1.63p-1f + (fe * -0x1.bd0106p-13f + lm);  // e * ln2 in two words
}
RT_HD float rt_sinf(float x) {  // |x| up to a few thousand
    float kf = floorf(x * 0x1.45f306p-1f + 0.5f);          // nearest multiple of pi/2
    int k = (int)kf;
    float r = x - kf * 0x1.92p+0f;                          // pi/2 in three words (Cody-Waite)
    r = r - kf * 0x1.fb4p-12f;
    r = r - kf * 0x1.4442d2p-24f;
    float r2 = r * r;
    float sp = r + r * (r2 * (-0x1.555556p-3f + r2 * (0x1.111112p-7f + r2 * (-0x1.a01a02p-13f + r2 * 0x1.71de3ap-19f))));
    float cp = 1.0f + r2 * (-0.5f + r2 * (0x1.555556p-5f + r2 * (-0x1.6c16c2p-10f + r2 * 0x1.a01a02p-16f)));
    float v = (k & 1) ? cp : sp;
    return (k & 2) ? -v : v;
}
RT_HD float rt_asin_poly(float z) {  // (asin(x) - x) / x for z = x^2 <= 1/4, rational form of fdlibm's float asin
    return z * (0x1.5554eap-3f + z * (-0x1.5e2774p-5f + z * -0x1.1ba6d6p-7f)) / (1.0f + z * -0x1.69cb5cp-1f);
}
RT_HD float rt_acosf(float x) {  // |x| <= 1
    float ax = x < 0.0f ? -x : x;
    if (ax <= 0.5f) {
        float r = rt_asin_poly(x * x);
        return 0x1.921fb6p+0f - (x + x * r);
    }
    float z = (1.0f - ax) * 0.5f;
    float s = sqrtf(z);
    float a = 2.0f * (s + s * rt_asin_poly(z));             // acos(|x|)
    return x < 0.0f ? 0x1.921fb6p+1f - a : a;
}
RT_HD float rt_atan2f(float y, float x) {
    float ax = x < 0.0f ? -x : x, ay = y < 0.0f ? -y : y;
    if (ax == 0.0f && ay == 0.0f) return 0.0f;
    bool swap = ay > ax;
    float t = swap ? ax / ay : ay / ax;                     // [0, 1]
    float base = 0.0f;
    if (t > 0x1.a8279ap-2f) { t = (t - 1.0f) / (t + 1.0f); base = 0x1.921fb6p-1f; }  // tan(pi/8): atan t = pi/4 + atan((t-1)/(t+1))
    float z = t * t;
    float p = z * (-0x1.555556p-2f + z * (0x1.99999ap-3f + z * (-0x1.24924ap-3f + z * (0x1.c71c72p-4f + z * (-0x1.745d18p-4f +
              z * (0x1.3b13b2p-4f + z * (-0x1.111112p-4f + z * 0x1.e1e1e2p-5f)))))));
    float a = base + (t + t * p);
    if (swap) a = 0x1.921fb6p+0f - a;
    if (x < 0.0f) a = 0x1.921fb6p+1f - a;
    return y < 0.0f ? -a : a;
}

// Ray, rt_engine/ray_data.cuh:8-15
struct Ray {
    f3 o, d;
    float time;
};
RT_HD f3 ray_at(const Ray& r, float t) { return r.o + r.d * t; }

// ---------------------------------------------------------------------------------------------
// RNG replacing cuRandom's per-pixel XORWOW state (utilities/cuda_utilities/cuRandom.cuh:10-41; 48 B/pixel
// of global-memory state read-modify-written through a reference, Renderer.cu:191).  It keeps cuRandom's
// shape — a sequential stream of uniforms in (0,1], consumed in the reference's order — but the stream
// belongs to ONE SAMPLE and is seeded counter-based, so nothing lives in memory:
//   state  = philox4x32-10(counter = (0, sample, pixel, stream), key = seed)       (4 registers)
//   next() = one xoshiro128++ step (Blackman & Vigna), u = ((word >> 8) + 1) * 2^-24, exact in fp32
// One Philox per sample (at regeneration, where every lane runs it together) and ~12 full-rate integer
// instructions per uniform keep the rejection loops of cuRandomInUnit / cuRandomOnUnit cheap.
// ---------------------------------------------------------------------------------------------
#define RT_STREAM_RENDER 0u
#define RT_STREAM_SCENE 0x5CE9E5u

RT_HD void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        // one 32x32->64 product per word (v_mad_u64_u32 on gfx950: half the multiplies of a mul_hi/mul_lo pair)
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

RT_HD uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }

struct Rng {
    uint32_t s0, s1, s2, s3;
    uint32_t draws;  // uniforms consumed (read only by the probes; dead code in the render kernels)
    RT_HD void init(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t stream) {
        uint32_t o[4];
        philox4x32_10(0u, sample, pixel, stream, (uint32_t)seed, (uint32_t)(seed >> 32), o);
        s0 = o[0]; s1 = o[1]; s2 = o[2]; s3 = o[3];
        if ((s0 | s1 | s2 | s3) == 0u) s0 = 1u;  // xoshiro's one forbidden state
        draws = 0;
    }
    // cuRandom::next(), cuRandom.cuh:21
    RT_HD float next() {
        uint32_t result = rotl32(s0 + s3, 7) + s0;
        uint32_t t = s1 << 9;
        s2 ^= s0;
        s3 ^= s1;
        s1 ^= s2;
        s0 ^= s3;
        s2 ^= t;
        s3 = rotl32(s3, 11);
        draws++;
        return (float)((result >> 8) + 1u) * 5.9604644775390625e-08f;
    }
    // next() * 2 - 1, the form the rejection loops consume (glm_utils.h:86,94), in 4 instructions instead of 6: with k = (word >> 8) + 1
    // in [1, 2^24], u = k 2^-24 and u * 2 - 1 = (k - 2^23) 2^-23 are both exact in fp32, so this IS the same float (also for k = 2^23: +0).
    RT_HD float next_signed() {
        uint32_t result = rotl32(s0 + s3, 7) + s0;
        uint32_t t = s1 << 9;
        s2 ^= s0;
        s3 ^= s1;
        s1 ^= s2;
        s0 ^= s3;
        s2 ^= t;
        s3 = rotl32(s3, 11);
        draws++;
        return (float)(int32_t)((result >> 8) + 1u - 0x800000u) * 1.1920928955078125e-07f;
    }
};
// glm::cuRandomInUnit<2>, utils:84-90
RT_HD void rng_in_unit2(Rng& g, float& ox, float& oy) {
    for (;;) {
        float x = g.next_signed();
        float y = g.next_signed();
        if (length2(x, y) < 1.0f) { ox = x; oy = y; return; }
    }
}
// glm::cuRandomOnUnit<3>, utils:92-98
RT_HD f3 rng_on_unit3(Rng& g) {
    // the normalisation (sqrt + division) sits AFTER the rejection loop: inside it a wave would execute it once per
    // iteration in which any lane accepts, i.e. ~6 times per call instead of once
    // `!near_zero(v) && length2(v) < 1` (utils:95): every component is a multiple of 2^-23 (next_signed), so v is near zero (all
    // |v_i| <= 1e-9) exactly when it IS zero, i.e. when length2(v) == 0 (a non-zero v has length2 >= 2^-46, no underflow): one sum, two compares
    f3 v;
    for (;;) {
        v.x = g.next_signed();
        v.y = g.next_signed();
        v.z = g.next_signed();
        const float l2 = length2(v);
        if (l2 > 0.0f && l2 < 1.0f) break;
    }
    return normalize(v);
}

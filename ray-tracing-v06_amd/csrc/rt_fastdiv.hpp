// rt_fastdiv.hpp — correctly rounded fp32 division in 5 instructions, for the BVH box tests.
//
// aabb::intersects (rt_engine/geometry/aabb.cuh:30-31) divides six plane offsets by the ray direction
// for EVERY box: (min - o) / d, (max - o) / d.  An IEEE-correct fp32 `/` costs ~11 instructions on
// gfx950 (v_div_scale x2, v_rcp, 4-5 v_fma, v_div_fmas, v_div_fixup), i.e. ~130 of the ~300
// instructions of one inner-node visit.  The direction is constant along a ray, so r = RN(1/d) is
// computed once per ray with a true division and each quotient is then recovered EXACTLY:
//
//      q0 = RN(n * r)                 |q0 - n/d| <= ~2 ulp          (two roundings)
//      e0 = fma(-q0, d, n)            residual of q0
//      q1 = fma(e0, r, q0)            faithful: |q1 - n/d| < 1 ulp
//      e1 = fma(-q1, d, n)            exact (q1 faithful)
//      q2 = fma(e1, r, q1)            == RN(n/d)   (Markstein 1990: r = RN(1/d), q1 faithful)
//
// The argument needs no overflow/underflow anywhere, which holds when (a) every box coordinate b of the
// scene is 0 or 2^-40 <= |b| < 2^40 (checked once on the host when the scene is packed) and (b) the ray is
// "regular" (ray_is_regular): 2^-40 <= |d_i| < 2^40, and o_i == 0 or 2^-60 <= |o_i| < 2^40.
// Then n = b - o is 0 or 2^-64 <= |n| < 2^41 (a non-zero difference of two floats is at least one ulp of
// the smaller), |n/d| is 0 or within [2^-104, 2^81], the residuals are multiples of 2^-111 or more, and
// every intermediate is a normal number or an exact zero.  Rays or scenes outside the class take the verbatim `/` path; the class has no NaN/inf, so IEEE
// min/max equal GLM's (y<x)?y:x up to the sign of a zero, which no comparison of the box test can see.
// tools/verify_fastdiv.hip checks q2 == n/d over EVERY pair of fp32 significands (2^46 pairs) on the
// GPU; tests/test_gpu_parity.py re-checks 2^32 random pairs and the box-test decisions each run.
#pragma once
#include "rt_math.hpp"

__device__ __forceinline__ float fast_div_exact(float n, float d, float r) {
    float q0 = n * r;
    float e0 = __builtin_fmaf(-q0, d, n);
    float q1 = __builtin_fmaf(e0, r, q0);
    float e1 = __builtin_fmaf(-q1, d, n);
    return __builtin_fmaf(e1, r, q1);
}

RT_HD bool coord_is_regular(float b) {  // b == 0 or 2^-40 <= |b| < 2^40
    uint32_t e;
#if defined(__HIP_DEVICE_COMPILE__)
    e = (__float_as_uint(b) >> 23) & 0xffu;
    bool zero = (__float_as_uint(b) << 1) == 0u;
#else
    uint32_t u;
    __builtin_memcpy(&u, &b, 4);
    e = (u >> 23) & 0xffu;
    bool zero = (u << 1) == 0u;
#endif
    return zero || (e >= 127u - 40u && e <= 127u + 39u);
}

__device__ __forceinline__ bool ray_is_regular(const Ray& ray) {
    uint32_t ok = 1u;
    const float dv[3] = {ray.d.x, ray.d.y, ray.d.z};
    const float ov[3] = {ray.o.x, ray.o.y, ray.o.z};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        uint32_t ed = (__float_as_uint(dv[k]) >> 23) & 0xffu;   // 2^-40 <= |d| < 2^40
        ok &= (ed >= 127u - 40u && ed <= 127u + 39u) ? 1u : 0u;
        uint32_t uo = __float_as_uint(ov[k]);
        uint32_t eo = (uo >> 23) & 0xffu;                        // o == 0 or 2^-60 <= |o| < 2^40
        ok &= ((uo << 1) == 0u || (eo >= 127u - 60u && eo <= 127u + 39u)) ? 1u : 0u;
    }
    return ok != 0u;
}

// The box test of aabb.cuh:30-44 on a regular ray: same decisions, same dist (up to the sign of zero).
__device__ __forceinline__ bool aabb_intersects_regular(f3 box_min, f3 box_max, const Ray& ray, f3 inv_d, float ray_max_dist, float& dist) {
    float ax = fast_div_exact(box_min.x - ray.o.x, ray.d.x, inv_d.x);
    float ay = fast_div_exact(box_min.y - ray.o.y, ray.d.y, inv_d.y);
    float az = fast_div_exact(box_min.z - ray.o.z, ray.d.z, inv_d.z);
    float bx = fast_div_exact(box_max.x - ray.o.x, ray.d.x, inv_d.x);
    float by = fast_div_exact(box_max.y - ray.o.y, ray.d.y, inv_d.y);
    float bz = fast_div_exact(box_max.z - ray.o.z, ray.d.z, inv_d.z);
    float tmin = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    float tmax = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    bool hit = tmin <= tmax && tmin < ray_max_dist && tmax > 0;
    if (hit) dist = tmin;
    return hit;
}

// rt_fastdiv.hpp — correctly rounded fp32 division in 5 (one reciprocal word) or 4 (two words) instructions, and the
// correctly rounded reciprocal in 3, for the BVH box tests.
//
// aabb::intersects (rt_engine/geometry/aabb.cuh:30-31) divides six plane offsets by the ray direction
// for EVERY box: (min - o) / d, (max - o) / d.  An IEEE-correct fp32 `/` costs ~11 instructions on
// gfx950 (v_div_scale x2, v_rcp, 4-5 v_fma, v_div_fmas, v_div_fixup), i.e. ~130 of the ~300
// instructions of one inner-node visit.  The direction is constant along a ray, so r = RN(1/d) is
// computed once per ray (rcp_exact_regular) and each quotient is then recovered EXACTLY (the hot loop uses the 4-instruction
// form fast_div_exact4 further down; this one serves the root box and the filtered variant):
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

// RN(1 / x) for 2^-40 <= |x| < 2^40: the hardware reciprocal approximation (1 ulp) plus ONE Newton step is the correctly
// rounded quotient for every such x — checked exhaustively over all 1,342,177,280 of them against the IEEE division by
// rt_selftest_fastrcp, run by the GPU tests (0 mismatches).  3 instructions instead of the ~11 of a true division, three times per ray.
__device__ __forceinline__ float rcp_exact_regular(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}

// The same quotient in 4 instructions, from a two-word reciprocal rh + rl ~= 1/d (rh = RN(1/d), rl = RN(rh * (1 - d*rh)),
// both per ray):
//      q0 = RN(n * rh)                  |q0 - n*rh| <= 1/2 ulp
//      q1 = fma(n, rl, q0)              = RN(n/d (1 + O(2^-47)) + (q0 - n*rh)): one of the two floats around n/d (faithful)
//      e1 = fma(-q1, d, n)              exact, since q1 is faithful
//      q2 = fma(e1, rh, q1)             == RN(n/d)   (Markstein: rh = RN(1/d), q1 faithful)
// Same class conditions as fast_div_exact; checked over every significand pair by rt_selftest_fastdiv (mode 1).
__device__ __forceinline__ float rcp_low_word(float d, float rh) { return __builtin_fmaf(-d, rh, 1.0f) * rh; }
__device__ __forceinline__ float fast_div_exact4(float n, float d, float rh, float rl) {
    float q0 = n * rh;
    float q1 = __builtin_fmaf(n, rl, q0);
    float e1 = __builtin_fmaf(-q1, d, n);
    return __builtin_fmaf(e1, rh, q1);
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
    // magnitudes as integers: for non-negative floats the bit patterns order like the values, inf / NaN are the largest
    const uint32_t dx = __float_as_uint(ray.d.x) & 0x7fffffffu, dy = __float_as_uint(ray.d.y) & 0x7fffffffu, dz = __float_as_uint(ray.d.z) & 0x7fffffffu;
    const uint32_t ox = __float_as_uint(ray.o.x) & 0x7fffffffu, oy = __float_as_uint(ray.o.y) & 0x7fffffffu, oz = __float_as_uint(ray.o.z) & 0x7fffffffu;
    const uint32_t lo_d = (127u - 40u) << 23, hi = (127u + 40u) << 23, lo_o = (127u - 60u) << 23;   // 2^-40, 2^40, 2^-60
    const uint32_t d_min = min(min(dx, dy), dz), d_max = max(max(dx, dy), dz);           // 2^-40 <= |d| < 2^40
    // o == 0 or 2^-60 <= |o| < 2^40: subtracting one sends the pattern of zero to the top, so one minimum covers both cases
    const uint32_t o_min = min(min(ox - 1u, oy - 1u), oz - 1u), o_max = max(max(ox, oy), oz);
    return d_min >= lo_d && d_max < hi && o_min >= lo_o - 1u && o_max < hi;
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


// The same test with the per-axis min/max resolved by the caller: (nx, ny, nz) are the box planes the ray ENTERS
// through and (fx, fy, fz) the ones it LEAVES through (near = box min where d > 0, box max where d < 0).  RN(n / d)
// is monotone in n, and b_min <= b_max gives RN(b_min - o) <= RN(b_max - o), so the near quotient IS min(t0, t1)
// and the far one max(t0, t1) of aabb.cuh:34-39 (up to the sign of a zero) — no v_min/v_max per axis.
__device__ __forceinline__ bool slab_near_far_regular(float nx, float ny, float nz, float fx, float fy, float fz,
                                                      const Ray& ray, f3 inv_d, f3 inv_lo, float ray_max_dist, float& tmin_out) {
    float tnx = fast_div_exact4(nx - ray.o.x, ray.d.x, inv_d.x, inv_lo.x);
    float tny = fast_div_exact4(ny - ray.o.y, ray.d.y, inv_d.y, inv_lo.y);
    float tnz = fast_div_exact4(nz - ray.o.z, ray.d.z, inv_d.z, inv_lo.z);
    float tfx = fast_div_exact4(fx - ray.o.x, ray.d.x, inv_d.x, inv_lo.x);
    float tfy = fast_div_exact4(fy - ray.o.y, ray.d.y, inv_d.y, inv_lo.y);
    float tfz = fast_div_exact4(fz - ray.o.z, ray.d.z, inv_d.z, inv_lo.z);
    float tmin = fmaxf(fmaxf(tnx, tny), tnz);
    float tmax = fminf(fminf(tfx, tfy), tfz);
    tmin_out = tmin;
    return tmin <= tmax && tmin < ray_max_dist && tmax > 0;
}

// CERTIFIED FAR PLANES (the default hot loop since round 4): what the traversal needs of the far planes of a box is ONE decision, `tmin <= tmax`
// (aabb.cuh:41), plus the sign of tmax.  The near parameters stay exact quotients — `tmin` is compared with the sibling's `tmin` (BVH.cu:90), where
// boxes that share a plane tie exactly and often — but the far ones are products with the rounded reciprocal, t' = RN(n * RN(1/d)) (2 instead of 5
// instructions per plane), and the decision taken with them is CERTIFIED:
//     |t' - RN(n/d)| <= 3.02 u |n/d|,  u = 2^-24   (reciprocal, product and quotient round once each; no under- or overflow in the class)
//  => |min3(t') - min3(q)| <= 3.1 u |min3(t')|
//  => if |min3(t') - tmin| > 8 u max(|min3(t')| of both boxes), then  tmin <= min3(q)  <=>  tmin <= min3(t').
// A lane that cannot certify one of its two boxes (a ray that grazes a box edge to within ~2^-21 of the parameter: ~1e-6 of the visits) makes the WAVE
// redo the far planes of the visit with exact quotients (a wave-uniform, rare branch).  The sign of t' is the sign of the quotient, and zero together
// with it, so `tmax > 0` needs no margin.  Same decisions as aabb::intersects on every ray of the class by construction; checked against the exact
// form on adversarial inputs (grazing rays, far parameters within a few ulp of the near ones) by rt_probe_boxpair_certified / tests/test_gpu_parity.py.
#define RT_FAR_EPS 4.76837158203125e-07f   /* 2^-21 = 8 u */
__device__ __forceinline__ float slab_near_exact(float nx, float ny, float nz, const Ray& ray, f3 inv_d, f3 inv_lo) {
    return fmaxf(fmaxf(fast_div_exact4(nx - ray.o.x, ray.d.x, inv_d.x, inv_lo.x), fast_div_exact4(ny - ray.o.y, ray.d.y, inv_d.y, inv_lo.y)),
                 fast_div_exact4(nz - ray.o.z, ray.d.z, inv_d.z, inv_lo.z));
}
__device__ __forceinline__ float slab_far_exact(float fx, float fy, float fz, const Ray& ray, f3 inv_d, f3 inv_lo) {
    return fminf(fminf(fast_div_exact4(fx - ray.o.x, ray.d.x, inv_d.x, inv_lo.x), fast_div_exact4(fy - ray.o.y, ray.d.y, inv_d.y, inv_lo.y)),
                 fast_div_exact4(fz - ray.o.z, ray.d.z, inv_d.z, inv_lo.z));
}
__device__ __forceinline__ float slab_far_product(float fx, float fy, float fz, const Ray& ray, f3 inv_d) {
    return fminf(fminf((fx - ray.o.x) * inv_d.x, (fy - ray.o.y) * inv_d.y), (fz - ray.o.z) * inv_d.z);
}
// true when the two `tmin <= tmax` decisions taken with the product forms (far_l, far_r) are NOT both certain
__device__ __forceinline__ bool far_pair_uncertain(float tmin_l, float far_l, float tmin_r, float far_r) {
    const float gap = fminf(fabsf(far_l - tmin_l), fabsf(far_r - tmin_r));
    const float scale = fmaxf(fabsf(far_l), fabsf(far_r));
    return gap <= scale * RT_FAR_EPS;
}

// The root box of a trace (BVH.cu:59-60) decides one thing — is the world entered at all, against a fresh rec.distance = _MISS_DIST — so ALL six of its
// plane parameters can be products: `tmin <= tmax` is certified as above (both sides are approximate now: twice the margin), `tmin < _MISS_DIST` holds
// for every parameter of the class, `tmax > 0` is sign-exact; a lane that cannot certify (a ray that grazes the world's bounds) takes the exact test.
__device__ __forceinline__ bool root_box_hit_certified(f3 box_min, f3 box_max, const Ray& ray, f3 inv_d) {
    const float ax = (box_min.x - ray.o.x) * inv_d.x, ay = (box_min.y - ray.o.y) * inv_d.y, az = (box_min.z - ray.o.z) * inv_d.z;
    const float bx = (box_max.x - ray.o.x) * inv_d.x, by = (box_max.y - ray.o.y) * inv_d.y, bz = (box_max.z - ray.o.z) * inv_d.z;
    const float tmin = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    const float tmax = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    if (fabsf(tmax - tmin) <= fmaxf(fabsf(tmin), fabsf(tmax)) * (2.0f * RT_FAR_EPS)) {   // rare, per lane
        float d;
        return aabb_intersects_regular(box_min, box_max, ray, inv_d, RT_MISS_DIST, d);
    }
    return tmin <= tmax && tmax > 0;
}

// TOLERANCE MODE (kernel variant 6, opt-in, NOT bit-exact by construction): the same slab test with the quotients of aabb.cuh:30-31 replaced
// by products with the correctly rounded reciprocal, t = RN((b - o) * RN(1/d)) — two roundings instead of one, at most ~1 ulp off the true quotient,
// and still a monotone function of the plane offset for a given ray (so planes that two boxes share still tie exactly).  What BASELINE.json's
// |delta| < 1e-3 allows and the oracle's bits do not promise.  Measured (round 4, EXPERIMENTS.md E4): 24 instead of 60 instructions for the twelve
// plane parameters of a visit, the dominant kernel 1.26x / 1.24x faster on BASELINE configs[1..2] with 0 differing pixels on the full-size frames.
// Sphere worlds only: with quads (whose edges coincide with their boxes' edges) the Cornell box at 5000 spp moved one pixel by 2.1e-3, so worlds
// beyond the reference's feature set are refused; the one-instruction form fma(b, RN(1/d), -o * RN(1/d)) moved pixels by up to 0.08 and was deleted.
__device__ __forceinline__ bool slab_near_far_tolerant(float nx, float ny, float nz, float fx, float fy, float fz,
                                                       const Ray& ray, f3 inv_d, float ray_max_dist, float& tmin_out) {
    const float tmin = fmaxf(fmaxf((nx - ray.o.x) * inv_d.x, (ny - ray.o.y) * inv_d.y), (nz - ray.o.z) * inv_d.z);
    const float tmax = fminf(fminf((fx - ray.o.x) * inv_d.x, (fy - ray.o.y) * inv_d.y), (fz - ray.o.z) * inv_d.z);
    tmin_out = tmin;
    return tmin <= tmax && tmin < ray_max_dist && tmax > 0;
}


// ---------------------------------------------------------------------------------------------------
// Filtered predicates for one inner-node visit (both child boxes) on a regular ray.
//
// The traversal needs only DECISIONS from the two box tests (BVH.cu:87-96): hit_left, hit_right, and whether
// left_dist > right_dist.  With t' = RN(n * r) (one multiply instead of the 5-instruction exact quotient)
// every plane parameter is known to |t' - t| <= 2^-22.9 |t'|  (r = RN(1/d) and the product each round once),
// and since x -> x +- delta|x| is monotone, the same relative bound carries over to tmin' = max3(min..) and
// tmax' = min3(max..).  A comparison whose operands are further apart than RT_FILTER_EPS = 2^-21 (4x the
// bound) relative to the larger one therefore has the same outcome as the exact comparison; signs of the
// t' are exact (n * r and n / d have the same sign and are zero together in the regular class), so
// `tmax > 0` needs no margin.
//
// Near-ties of the two entry distances are NOT rare: sibling boxes share planes (every small sphere of the
// book scenes spans y in [0, 0.4]), and 1.6 % of visits have both children hit at exactly the same tmin
// through the same plane (all near-ties measured are of this kind).  They are settled without division:
// if, for both boxes, exactly one axis has its entry parameter within the margin of tmin' (so that axis IS
// the exact argmax), it is the same axis, and the two entry plane coordinates are equal, then the exact
// quotients are the same number: left_dist == right_dist, no swap.  Anything else near a margin (~1e-6 of
// visits) is reported `uncertain` and the caller redoes the visit with exact quotients.
// Needs non-inverted boxes (bmin <= bmax: the entry plane of axis k is bmin_k iff d_k > 0), checked at pack time.
// ---------------------------------------------------------------------------------------------------
#define RT_FILTER_EPS 4.76837158203125e-07f  // 2^-21

struct BoxPairDecision {
    bool hit_left, hit_right, swap, uncertain;
    uint32_t why;  // diagnostics: bit0 left hit unsure, bit1 right hit unsure, bit2 order near-tie, bit3 tie not settled by the plane rule
};

struct SlabApprox {
    float ex, ey, ez;  // per-axis entry parameters
    float tmin, tmax;
};

__device__ __forceinline__ SlabApprox slab_approx(f3 bmin, f3 bmax, const Ray& ray, f3 inv_d) {
    float ax = (bmin.x - ray.o.x) * inv_d.x, bx = (bmax.x - ray.o.x) * inv_d.x;
    float ay = (bmin.y - ray.o.y) * inv_d.y, by = (bmax.y - ray.o.y) * inv_d.y;
    float az = (bmin.z - ray.o.z) * inv_d.z, bz = (bmax.z - ray.o.z) * inv_d.z;
    SlabApprox s;
    s.ex = fminf(ax, bx); s.ey = fminf(ay, by); s.ez = fminf(az, bz);
    s.tmin = fmaxf(fmaxf(s.ex, s.ey), s.ez);
    s.tmax = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    return s;
}

// hit = tmin <= tmax && tmin < maxd && tmax > 0 (aabb.cuh:41) as (certainly true, certainly false) pair
__device__ __forceinline__ void hit_filtered(float tmin, float tmax, float maxd, bool& yes, bool& no) {
    float m1 = RT_FILTER_EPS * fmaxf(fabsf(tmin), fabsf(tmax));
    bool c1_yes = tmin <= tmax - m1, c1_no = tmin > tmax + m1;
    float m2 = RT_FILTER_EPS * fabsf(tmin);
    bool c2_yes = tmin < maxd - m2, c2_no = tmin >= maxd + m2;
    bool c3 = tmax > 0;
    yes = c1_yes && c2_yes && c3;
    no = c1_no || c2_no || !c3;
}

__device__ __forceinline__ BoxPairDecision box_pair_filtered(f3 lmin, f3 lmax, f3 rmin, f3 rmax, const Ray& ray, f3 inv_d, float maxd) {
    SlabApprox L = slab_approx(lmin, lmax, ray, inv_d);
    SlabApprox R = slab_approx(rmin, rmax, ray, inv_d);
    bool ly, ln, ry, rn;
    hit_filtered(L.tmin, L.tmax, maxd, ly, ln);
    hit_filtered(R.tmin, R.tmax, maxd, ry, rn);
    BoxPairDecision d;
    d.hit_left = ly;
    d.hit_right = ry;
    d.uncertain = !(ly || ln) || !(ry || rn);
    d.why = (!(ly || ln) ? 1u : 0u) | (!(ry || rn) ? 2u : 0u);
    // left_dist > right_dist (BVH.cu:90): a missed box keeps _MISS_DIST
    d.swap = !ly && ry;  // _MISS_DIST > right_dist; (hit, miss) and (miss, miss) do not swap
    if (ly && ry) {
        float m = RT_FILTER_EPS * fmaxf(fabsf(L.tmin), fabsf(R.tmin));
        d.swap = L.tmin > R.tmin + m;
        if (!d.swap && !(L.tmin < R.tmin - m)) {  // near-tie
            d.why |= 4u;
            const float thl = L.tmin - m, thr = R.tmin - m;
            const bool lx = L.ex >= thl, lyy = L.ey >= thl, lz = L.ez >= thl;
            const bool rx = R.ex >= thr, ryy = R.ey >= thr, rz = R.ez >= thr;
            const bool px = (ray.d.x > 0.0f ? lmin.x : lmax.x) == (ray.d.x > 0.0f ? rmin.x : rmax.x);
            const bool py = (ray.d.y > 0.0f ? lmin.y : lmax.y) == (ray.d.y > 0.0f ? rmin.y : rmax.y);
            const bool pz = (ray.d.z > 0.0f ? lmin.z : lmax.z) == (ray.d.z > 0.0f ? rmin.z : rmax.z);
            const bool unique = ((int)lx + (int)lyy + (int)lz == 1) && ((int)rx + (int)ryy + (int)rz == 1);
            const bool same_plane_tie = unique && ((lx && rx && px) || (lyy && ryy && py) || (lz && rz && pz));
            if (!same_plane_tie) {  // not seen in practice: let the caller redo the visit with exact quotients
                d.why |= 8u;
                d.uncertain = true;
            }
        }
    }
    return d;
}

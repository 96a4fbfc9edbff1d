/*
 * rt_oracle.c — CPU ORACLE (test infrastructure, NOT product code).
 * See rt_oracle.h for the pin status (pinned to the reference: the GLM vocabulary, aabb, HittableList /
 * bvh_node traversal, checker_texture, Ray::at / isBackfacing; "parity unpinned": the sphere test, the
 * flat BVH, Scatter, cameras, sample_world, render_kernel) and the arithmetic contract.  Build: oracle/Makefile
 * (gcc -O2 -ffp-contract=off, no fast-math).
 *
 * Citations are file:line under /root/reference/ ; "…/geometry" =
 * main/src/rt_engine/geometry, "…/shaders" = main/src/rt_engine/shaders,
 * "glm/" = Libraries/include/glm.
 */
#define _GNU_SOURCE   /* pthread_setaffinity_np, CPU_SET */
#include "rt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MISS_DIST 3.402823466e+38F /* _MISS_DIST, rt_engine/ray_data.cuh:17 */
#define ORC_STACK 32                   /* _PRIO_QUEUE_ELEM_COUNT, …/geometry/BVH.cu:17 */
#define ORC_PRIM_MOVING 0x80000000u

/* ------------------------------------------------------------------ */
/* GLM vocabulary                                                      */
/* ------------------------------------------------------------------ */
typedef orc_v3 v3;

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 ld3(const float* p) { return V(p[0], p[1], p[2]); }
static inline void st3(float* p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 divv(v3 a, v3 b) { return V(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline v3 muls(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 divs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
static inline v3 neg(v3 a) { return V(-a.x, -a.y, -a.z); }

/* glm/detail/func_common.inl:17-30: min = (y<x)?y:x ; max = (x<y)?y:x */
static inline float gmin(float x, float y) { return (y < x) ? y : x; }
static inline float gmax(float x, float y) { return (x < y) ? y : x; }
static inline v3 vmin(v3 a, v3 b) { return V(gmin(a.x, b.x), gmin(a.y, b.y), gmin(a.z, b.z)); }
static inline v3 vmax(v3 a, v3 b) { return V(gmax(a.x, b.x), gmax(a.y, b.y), gmax(a.z, b.z)); }
/* glm/gtx/component_wise.inl:111-126 */
static inline float comp_min(v3 a) { float r = a.x; r = gmin(r, a.y); r = gmin(r, a.z); return r; }
static inline float comp_max(v3 a) { float r = a.x; r = gmax(r, a.y); r = gmax(r, a.z); return r; }
/* glm/detail/func_geometric.inl:48-56: tmp = a*b; tmp.x + tmp.y + tmp.z */
static inline float dot(v3 a, v3 b) { v3 t = mul(a, b); return t.x + t.y + t.z; }
/* glm/detail/func_geometric.inl:70-81 */
static inline v3 cross(v3 x, v3 y) {
    return V(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
/* glm/detail/func_geometric.inl:84-93 + func_exponential.inl:134-139:
 * v * inversesqrt(dot(v,v)), inversesqrt(x) = 1 / sqrt(x) */
static inline v3 normalize(v3 a) { float inv = 1.0f / sqrtf(dot(a, a)); return muls(a, inv); }
/* glm/detail/func_geometric.inl:104-110: I - N * dot(N,I) * 2 */
static inline v3 reflect(v3 i, v3 n) { return sub(i, muls(muls(n, dot(n, i)), 2.0f)); }
/* glm/detail/func_geometric.inl:113-123 */
static inline v3 refract(v3 i, v3 n, float eta) {
    float dv = dot(n, i);
    float k = 1.0f - eta * eta * (1.0f - dv * dv);
    if (k >= 0.0f) return sub(muls(i, eta), muls(n, eta * dv + sqrtf(k)));
    return V(0.0f, 0.0f, 0.0f);
}
/* glm/detail/func_common.inl:104-112,124-132: x*(1-a) + y*a */
static inline v3 mix3(v3 x, v3 y, float a) { return add(muls(x, 1.0f - a), muls(y, a)); }
static inline float mix1(float x, float y, float a) { return x * (1.0f - a) + y * a; }
/* main/src/utilities/glm_utils.h:15-25, epsilon 1e-9f */
static inline int near_zero(v3 a) {
    if (fabsf(a.x) > 1e-9f) return 0;
    if (fabsf(a.y) > 1e-9f) return 0;
    if (fabsf(a.z) > 1e-9f) return 0;
    return 1;
}
/* glm_utils.h:27-35: sum{} ; sum += v[i]*v[i] */
static inline float length2_3(v3 a) { float s = 0.0f; s += a.x * a.x; s += a.y * a.y; s += a.z * a.z; return s; }
static inline float length2_2(float x, float y) { float s = 0.0f; s += x * x; s += y * y; return s; }
/* glm_utils.h:64-67: a + (b - a) * f */
static inline v3 lerp3(v3 a, v3 b, float f) { return add(a, muls(sub(b, a), f)); }
/* glm/detail/func_trigonometric.inl:9-14 */
static inline float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }

float orc_glm_dot(const float a[3], const float b[3]) { return dot(ld3(a), ld3(b)); }
void orc_glm_cross(const float a[3], const float b[3], float out[3]) { st3(out, cross(ld3(a), ld3(b))); }
void orc_glm_normalize(const float a[3], float out[3]) { st3(out, normalize(ld3(a))); }
void orc_glm_reflect(const float i[3], const float n[3], float out[3]) { st3(out, reflect(ld3(i), ld3(n))); }
void orc_glm_refract(const float i[3], const float n[3], float eta, float out[3]) { st3(out, refract(ld3(i), ld3(n), eta)); }
void orc_glm_mix3(const float a[3], const float b[3], float t, float out[3]) { st3(out, mix3(ld3(a), ld3(b), t)); }
float orc_glm_mix1(float a, float b, float t) { return mix1(a, b, t); }
void orc_glm_min3(const float a[3], const float b[3], float out[3]) { st3(out, vmin(ld3(a), ld3(b))); }
void orc_glm_max3(const float a[3], const float b[3], float out[3]) { st3(out, vmax(ld3(a), ld3(b))); }
float orc_glm_compmax(const float a[3]) { return comp_max(ld3(a)); }
float orc_glm_compmin(const float a[3]) { return comp_min(ld3(a)); }
int orc_glm_near_zero(const float a[3]) { return near_zero(ld3(a)); }
float orc_glm_length2(const float a[3]) { return length2_3(ld3(a)); }
void orc_glm_lerp(const float a[3], const float b[3], float t, float out[3]) { st3(out, lerp3(ld3(a), ld3(b), t)); }
float orc_glm_radians(float deg) { return radians(deg); }
static inline uint32_t f2u(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
static inline float u2f(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }
/* ---------------------------------------------------------------------------------------------
 * Deterministic fp32 elementary functions for the extension materials (constant media, Perlin marble, sphere uv).
 * The reference has none of these features; what matters here is that this CPU oracle and the GPU compute the SAME
 * bits, which libm / ocml do not promise.  So: plain fp32 +, -, *, /, sqrt, floor and integer bit operations only,
 * in one fixed order, no fused multiply-add (both sides build with -ffp-contract=off).  Accuracy is a few ulp.
 * --------------------------------------------------------------------------------------------- */
static float m_logf(float x) {  // x > 0, finite, normal
    uint32_t ix = f2u(x);
    int e = (int)(ix >> 23) - 127;
    float m = u2f((ix & 0x007fffffu) | 0x3f800000u);  // [1, 2)
    if (m > 0x1.6a09e6p+0f) { m = m * 0.5f; e += 1; }     // [sqrt(1/2), sqrt(2))
    float f = m - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    float p = z * (0x1.555556p-1f + z * (0x1.99999ap-2f + z * (0x1.24924ap-2f + z * 0x1.c71c72p-3f)));  // 2/3, 2/5, 2/7, 2/9
    float lm = s * (2.0f + p);                             // log(m) = 2 atanh(s)
    float fe = (float)e;
    return fe * 0x1.63p-1f + (fe * -0x1.bd0106p-13f + lm);  // e * ln2 in two words
}
static float m_sinf(float x) {  // |x| up to a few thousand
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
static float m_asin_poly(float z) {  // (asin(x) - x) / x for z = x^2 <= 1/4, rational form of fdlibm's float asin
    return z * (0x1.5554eap-3f + z * (-0x1.5e2774p-5f + z * -0x1.1ba6d6p-7f)) / (1.0f + z * -0x1.69cb5cp-1f);
}
static float m_acosf(float x) {  // |x| <= 1
    float ax = x < 0.0f ? -x : x;
    if (ax <= 0.5f) {
        float r = m_asin_poly(x * x);
        return 0x1.921fb6p+0f - (x + x * r);
    }
    float z = (1.0f - ax) * 0.5f;
    float s = sqrtf(z);
    float a = 2.0f * (s + s * m_asin_poly(z));             // acos(|x|)
    return x < 0.0f ? 0x1.921fb6p+1f - a : a;
}
static float m_atan2f(float y, float x) {
    float ax = x < 0.0f ? -x : x, ay = y < 0.0f ? -y : y;
    if (ax == 0.0f && ay == 0.0f) return 0.0f;
    int swap = ay > ax;
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
float orc_math_log(float x) { return m_logf(x); }
float orc_math_sin(float x) { return m_sinf(x); }
float orc_math_acos(float x) { return m_acosf(x); }
float orc_math_atan2(float y, float x) { return m_atan2f(y, x); }
/* fn: 0 log(a), 1 sin(a), 2 acos(a), 3 atan2(a, b) */
void orc_math_batch(int fn, size_t n, const float* a, const float* b, float* out) {
    for (size_t i = 0; i < n; i++)
        out[i] = fn == 0 ? m_logf(a[i]) : fn == 1 ? m_sinf(a[i]) : fn == 2 ? m_acosf(a[i]) : m_atan2f(a[i], b[i]);
}

/* main/src/Renderer.cu:209-211: clamp(x,0,1) = min(max(x,0),1)
 * (glm/detail/func_common.inl:240-246), then sqrt */
static inline v3 clamp01_sqrt(v3 a) {
    v3 c = vmin(vmax(a, V(0.0f, 0.0f, 0.0f)), V(1.0f, 1.0f, 1.0f));
    return V(sqrtf(c.x), sqrtf(c.y), sqrtf(c.z));
}
void orc_glm_clamp01_sqrt(const float a[3], float out[3]) { st3(out, clamp01_sqrt(ld3(a))); }

/* ------------------------------------------------------------------ */
/* RNG — the build's own generator, replacing cuRandom                  */
/* (utilities/cuda_utilities/cuRandom.cuh:10-41): it keeps cuRandom's    */
/* shape — a sequential stream of uniforms in (0,1] consumed in the      */
/* reference's order — but the stream belongs to ONE SAMPLE and is       */
/* seeded counter-based, so no state lives in memory:                    */
/*   state  = philox4x32-10(counter = (0, sample, pixel, stream),        */
/*                          key = seed lo/hi)        (128 bits)          */
/*   next() = xoshiro128++ step on that state (Blackman & Vigna),        */
/*            u = ((word >> 8) + 1) * 2^-24          (exact in fp32)     */
/* ------------------------------------------------------------------ */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

typedef struct {
    uint32_t s[4];
    uint32_t draws; /* uniforms consumed so far */
} rng_t;

static inline void rng_init(rng_t* g, uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t stream) {
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t ctr[4] = {0u, sample, pixel, stream};
    orc_philox4x32_10(ctr, key, g->s);
    if ((g->s[0] | g->s[1] | g->s[2] | g->s[3]) == 0u) g->s[0] = 1u; /* xoshiro's one forbidden state */
    g->draws = 0;
}
static inline uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
static inline uint32_t xoshiro128pp(uint32_t s[4]) {
    uint32_t result = rotl32(s[0] + s[3], 7) + s[0];
    uint32_t t = s[1] << 9;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl32(s[3], 11);
    return result;
}
/* cuRandom::next(), cuRandom.cuh:21 (curand_uniform in (0,1]) */
static inline float rng_next(rng_t* g) {
    uint32_t w = xoshiro128pp(g->s);
    g->draws++;
    return (float)((w >> 8) + 1u) * 5.9604644775390625e-08f; /* 2^-24 */
}
void orc_rng_uniforms(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t stream, uint32_t n, float* out) {
    rng_t g; rng_init(&g, seed, pixel, sample, stream);
    for (uint32_t i = 0; i < n; i++) out[i] = rng_next(&g);
}
/* glm::cuRandomInUnit<2>, utilities/glm_utils.h:84-90 */
static inline void rng_in_unit2(rng_t* g, float* ox, float* oy) {
    for (;;) {
        float x = rng_next(g) * 2.0f - 1.0f;
        float y = rng_next(g) * 2.0f - 1.0f;
        if (length2_2(x, y) < 1.0f) { *ox = x; *oy = y; return; }
    }
}
/* glm::cuRandomOnUnit<3>, utilities/glm_utils.h:92-98 */
static inline v3 rng_on_unit3(rng_t* g) {
    for (;;) {
        v3 v;
        v.x = rng_next(g) * 2.0f - 1.0f;
        v.y = rng_next(g) * 2.0f - 1.0f;
        v.z = rng_next(g) * 2.0f - 1.0f;
        if (!near_zero(v) && length2_3(v) < 1.0f) return normalize(v);
    }
}

/* ------------------------------------------------------------------ */
/* rays, boxes, spheres                                                */
/* ------------------------------------------------------------------ */
typedef struct { v3 o, d; float time; } ray_t;                 /* Ray, rt_engine/ray_data.cuh:8-15 */
typedef struct { float distance; v3 normal; int32_t prim; uint32_t mat; } rec_t; /* RayPayload + TraceRecord, ray_data.cuh:33-40, SphereHittable.cuh:38-41 */

static inline v3 ray_at(const ray_t* r, float t) { return add(r->o, muls(r->d, t)); } /* ray_data.cuh:14 */

/* aabb::intersects, …/geometry/aabb.cuh:30-44 */
static inline int aabb_intersects(v3 bmin_c, v3 bmax_c, const ray_t* ray, float ray_max_dist, float* dist) {
    v3 bmin = divv(sub(bmin_c, ray->o), ray->d);
    v3 bmax = divv(sub(bmax_c, ray->o), ray->d);
    v3 tmp_min = vmin(bmin, bmax);
    bmax = vmax(bmin, bmax);
    bmin = tmp_min;
    float tmin = comp_max(bmin);
    float tmax = comp_min(bmax);
    int hit = tmin <= tmax && tmin < ray_max_dist && tmax > 0;
    if (hit) *dist = tmin;
    return hit;
}

/* _sphere_closest_intersection, …/geometry/SphereHittable.cuh:15-33 */
static inline float sphere_closest_intersection(const ray_t* ray, v3 center, float radius) {
    v3 oc = sub(ray->o, center);
    float a = dot(ray->d, ray->d);
    float hb = dot(ray->d, oc);
    float c = dot(oc, oc) - radius * radius;
    float d = hb * hb - a * c;
    if (d <= 0) return ORC_MISS_DIST;
    d = sqrtf(d);
    float t = (-hb - d) / a;
    if (t < 0.0f) {
        t = (-hb + d) / a;
        if (t < 0.0f) return ORC_MISS_DIST;
    }
    return t;
}

/* SphereHittable::ClosestIntersection (…/geometry/SphereHittable.cu:56-66) and
 * MovingSphereHittable::ClosestIntersection (:91-102) */
static inline int prim_closest_intersection(const orc_world* w, int32_t idx, const ray_t* ray, rec_t* rec,
                                            orc_counters* cnt, rng_t* g) {
    const orc_prim* p = &w->prims[idx];
    v3 center = ld3(p->c0);
    if (p->mat & ORC_PRIM_MOVING) center = mix3(ld3(p->c0), ld3(p->c1), ray->time);
    cnt->leaf_tests++;
    const orc_material* pm = &w->materials[p->mat & ~ORC_PRIM_MOVING];
    if (pm->type == 5) { /* RT_MAT_ISOTROPIC */
        /* constant_medium::hit of "The Next Week" (extension; a sphere whose material is isotropic IS a constant medium
         * bounded by that sphere, density = param), in the reference's conventions: the ray interval is [0, rec.distance)
         * instead of [t_min, t_max], a tangent ray misses (SphereHittable.cuh:22).  ONE uniform is drawn per test that
         * finds a non-empty interval inside the boundary, in traversal order. */
        v3 oc = sub(ray->o, center);
        float a = dot(ray->d, ray->d);
        float hb = dot(ray->d, oc);
        float c = dot(oc, oc) - p->radius * p->radius;
        float d = hb * hb - a * c;
        if (d <= 0) return 0;
        d = sqrtf(d);
        float t1 = (-hb - d) / a, t2 = (-hb + d) / a;
        float t_in = t1 < 0.0f ? 0.0f : t1;
        float t_out = t2 > rec->distance ? rec->distance : t2;
        if (!(t_in < t_out)) return 0;
        float ray_length = sqrtf(a);
        float distance_inside = (t_out - t_in) * ray_length;
        float hit_distance = (-1.0f / pm->param) * m_logf(rng_next(g));
        if (hit_distance > distance_inside) return 0;
        float tm = t_in + hit_distance / ray_length;
        if (tm >= rec->distance) return 0;
        rec->mat = p->mat & ~ORC_PRIM_MOVING;
        rec->distance = tm;
        rec->prim = idx;
        rec->normal = V(1.0f, 0.0f, 0.0f);   /* arbitrary, as in the book: an isotropic scatter ignores it */
        return 1;
    }
    float t = sphere_closest_intersection(ray, center, p->radius);
    if (t >= rec->distance) return 0;
    rec->mat = p->mat & ~ORC_PRIM_MOVING;
    rec->distance = t;
    rec->prim = idx;
    rec->normal = divs(sub(ray_at(ray, rec->distance), center), p->radius);
    return 1;
}

/* quad::hit of "Ray Tracing: The Next Week" in the reference's conventions: t >= 0 is accepted (the reference
 * has no t_min and offsets the origin instead, Renderer.cu:175), `t >= rec.distance` rejects; the stored normal
 * faces AGAINST the ray (the book's set_face_normal), so a quad is two-sided. */
static inline int quad_closest_intersection(const orc_world* w, int32_t qi, int32_t unified, const ray_t* ray, rec_t* rec, orc_counters* cnt) {
    const orc_quad* q = &w->quads[qi];
    cnt->leaf_tests++;
    v3 n = ld3(q->normal);
    float denom = dot(n, ray->d);
    if (fabsf(denom) < 1e-8f) return 0;
    float t = (q->D - dot(n, ray->o)) / denom;
    if (t < 0.0f) return 0;
    if (t >= rec->distance) return 0;
    v3 planar = sub(ray_at(ray, t), ld3(q->Q));
    float alpha = dot(ld3(q->w), cross(planar, ld3(q->v)));
    float beta = dot(ld3(q->w), cross(ld3(q->u), planar));
    if (!(alpha >= 0.0f && alpha <= 1.0f && beta >= 0.0f && beta <= 1.0f)) return 0;
    rec->mat = q->mat;
    rec->distance = t;
    rec->prim = unified;
    rec->normal = (dot(ray->d, n) > 0) ? neg(n) : n;
    return 1;
}

static inline int any_prim_closest_intersection(const orc_world* w, int32_t idx, const ray_t* ray, rec_t* rec, orc_counters* cnt, rng_t* g) {
    if ((uint32_t)idx >= w->n_prims) return quad_closest_intersection(w, idx - (int32_t)w->n_prims, idx, ray, rec, cnt);
    return prim_closest_intersection(w, idx, ray, rec, cnt, g);
}

static inline int node_box(const orc_node* n, const ray_t* ray, float maxd, float* dist, orc_counters* cnt) {
    cnt->box_tests++;
    return aabb_intersects(ld3(n->min), ld3(n->max), ray, maxd, dist);
}

/* BVH::ClosestIntersection, …/geometry/BVH.cu:54-106 (non-priority-queue branch) */
static int bvh_closest_intersection(const orc_world* w, const ray_t* ray, rec_t* rec, orc_counters* cnt, int* err, rng_t* g) {
    int32_t stack[ORC_STACK];
    int head = 0;
    const orc_node* nodes = w->nodes;
    float root_dist;
    if (!node_box(&nodes[w->root], ray, rec->distance, &root_dist, cnt)) return 0;
    stack[head++] = w->root;
    if ((uint32_t)head > cnt->max_stack) cnt->max_stack = head;
    int hit_any = 0;
    while (head != 0) {
        int32_t idx = stack[--head];
        const orc_node* node = &nodes[idx];
        if (node->left == -1) {
            hit_any |= any_prim_closest_intersection(w, node->right, ray, rec, cnt, g);
            continue;
        }
        float left_dist = ORC_MISS_DIST, right_dist = ORC_MISS_DIST;
        int32_t left_idx = node->left, right_idx = node->right;
        node_box(&nodes[left_idx], ray, rec->distance, &left_dist, cnt);
        node_box(&nodes[right_idx], ray, rec->distance, &right_dist, cnt);
        if (left_dist > right_dist) {
            int32_t ti = left_idx; left_idx = right_idx; right_idx = ti;
            float tf = left_dist; left_dist = right_dist; right_dist = tf;
        }
        if (head + 2 > ORC_STACK) { *err = 4; return hit_any; }
        if (right_dist < rec->distance) stack[head++] = right_idx;
        if (left_dist < rec->distance) stack[head++] = left_idx;
        if ((uint32_t)head > cnt->max_stack) cnt->max_stack = head;
    }
    return hit_any;
}

/* BVH::ClosestIntersection with the distance-sorted queue the reference carries but disables (`_USE_PRIO_QUEUE false`,
 * …/geometry/BVH.cu:17-49 and the `#if _USE_PRIO_QUEUE` branch :80-86): every child box that is hit is enqueued with its entry
 * distance, the queue is kept sorted (largest distance at the bottom) and the NEAREST entry of the whole frontier is
 * dequeued next — best-first instead of depth-first.  Culling is still at enqueue time only (the box test's
 * `tmin < rec.distance`); nothing is re-checked at dequeue.
 * The reference's enqueue is off by one — it increments `head` BEFORE writing `distances[head]` (BVH.cu:37-39), so the
 * distance lands one slot above its index and the first comparison reads an unwritten slot.  FIXED here, deliberately
 * and documented: index and distance are written to the same slot, then the insertion sort runs as written
 * (swap downwards while the entry above is farther, stop at the first that is not: equal distances keep arrival order
 * relative to the stop rule `>`).  Capacity 32 = _PRIO_QUEUE_ELEM_COUNT; the reference does not check it, this reports
 * error 4.  Selected per world (orc_world.traversal == 1): an alternative traversal, NOT the live path — images may
 * differ from the stack traversal in rounding near-ties, like those of another tree. */
#define ORC_QUEUE 32
static int bvh_closest_intersection_queue(const orc_world* w, const ray_t* ray, rec_t* rec, orc_counters* cnt, int* err, rng_t* g) {
    int32_t indices[ORC_QUEUE];
    float distances[ORC_QUEUE];
    int head = 0;
    const orc_node* nodes = w->nodes;
    float root_dist;
    if (!node_box(&nodes[w->root], ray, rec->distance, &root_dist, cnt)) return 0;
#define ORC_ENQUEUE(idx_, dist_)                                                                  \
    do {                                                                                          \
        if (head >= ORC_QUEUE) { *err = 4; return hit_any; }                                      \
        indices[head] = (idx_); distances[head] = (dist_); head++;                                \
        for (int i_ = head - 1; i_ >= 1; i_--) {                                                  \
            if (distances[i_] > distances[i_ - 1]) {                                              \
                float td_ = distances[i_]; distances[i_] = distances[i_ - 1]; distances[i_ - 1] = td_;   \
                int32_t ti_ = indices[i_]; indices[i_] = indices[i_ - 1]; indices[i_ - 1] = ti_;  \
            } else break;                                                                         \
        }                                                                                         \
        if ((uint32_t)head > cnt->max_stack) cnt->max_stack = head;                               \
    } while (0)
    int hit_any = 0;
    ORC_ENQUEUE(w->root, root_dist);
    while (head != 0) {
        int32_t idx = indices[--head];
        const orc_node* node = &nodes[idx];
        if (node->left == -1) {
            hit_any |= any_prim_closest_intersection(w, node->right, ray, rec, cnt, g);
            continue;
        }
        float left_dist = ORC_MISS_DIST, right_dist = ORC_MISS_DIST;
        if (node_box(&nodes[node->left], ray, rec->distance, &left_dist, cnt)) ORC_ENQUEUE(node->left, left_dist);
        if (node_box(&nodes[node->right], ray, rec->distance, &right_dist, cnt)) ORC_ENQUEUE(node->right, right_dist);
    }
#undef ORC_ENQUEUE
    return hit_any;
}

/* A 4-WIDE walk of the same binary tree ("wider (BVH4/8) nodes"; the reference has binary nodes only, BVH.cuh:16-25):
 * orc_world.traversal == 2.  A visit of an inner node looks TWO levels down: its candidates are the children of its children (a child that is
 * a leaf stands for itself), i.e. up to four boxes; each is tested against rec.distance as it is now (BVH.cu:87-88's rule), the candidates are
 * ordered nearest first (stable insertion sort on the entry distances, a missed box keeps _MISS_DIST) and pushed far-to-near iff
 * `dist < rec.distance` (BVH.cu:95-96: culling at push time only).  The intermediate children's own boxes are never tested.  An alternative
 * traversal, NOT the live path: like the queue it may differ from the stack walk in rounding near-ties.  The stack holds 32 entries
 * (BVH.cu:17); a visit pushes up to four, so a tree of depth d needs at most 3 * ceil(d / 2) + 1: error 4 beyond. */
static int bvh_closest_intersection_wide4(const orc_world* w, const ray_t* ray, rec_t* rec, orc_counters* cnt, int* err, rng_t* g) {
    int32_t stack[ORC_STACK];
    int head = 0;
    const orc_node* nodes = w->nodes;
    float root_dist;
    if (!node_box(&nodes[w->root], ray, rec->distance, &root_dist, cnt)) return 0;
    stack[head++] = w->root;
    if ((uint32_t)head > cnt->max_stack) cnt->max_stack = head;
    int hit_any = 0;
    while (head != 0) {
        const int32_t idx = stack[--head];
        const orc_node* node = &nodes[idx];
        if (node->left == -1) {
            hit_any |= any_prim_closest_intersection(w, node->right, ray, rec, cnt, g);
            continue;
        }
        int32_t cand[4];
        float dist[4];
        int n = 0;
        const int32_t kids[2] = {node->left, node->right};
        for (int k = 0; k < 2; k++) {
            const orc_node* c = &nodes[kids[k]];
            if (c->left == -1) cand[n++] = kids[k];
            else { cand[n++] = c->left; cand[n++] = c->right; }
        }
        for (int i = 0; i < n; i++) {
            dist[i] = ORC_MISS_DIST;
            node_box(&nodes[cand[i]], ray, rec->distance, &dist[i], cnt);
        }
        for (int i = 1; i < n; i++)   /* nearest first; equal distances keep their order */
            for (int j = i; j >= 1 && dist[j - 1] > dist[j]; j--) {
                float td = dist[j]; dist[j] = dist[j - 1]; dist[j - 1] = td;
                int32_t ti = cand[j]; cand[j] = cand[j - 1]; cand[j - 1] = ti;
            }
        if (head + n > ORC_STACK) { *err = 4; return hit_any; }
        for (int i = n - 1; i >= 0; i--)
            if (dist[i] < rec->distance) stack[head++] = cand[i];
        if ((uint32_t)head > cnt->max_stack) cnt->max_stack = head;
    }
    return hit_any;
}

/* HittableList::ClosestIntersection, …/geometry/HittableList.cuh:21-34 */
static int list_closest_intersection(const orc_world* w, const ray_t* ray, rec_t* rec, orc_counters* cnt, rng_t* g) {
    float d;
    cnt->box_tests++;
    if (!aabb_intersects(ld3(w->bounds_min), ld3(w->bounds_max), ray, rec->distance, &d)) return 0;
    int hit_any = 0;
    for (uint32_t i = 0; i < w->n_prims + w->n_quads; i++)
        if (any_prim_closest_intersection(w, (int32_t)i, ray, rec, cnt, g)) hit_any = 1;
    return hit_any;
}

/* bvh_node::ClosestIntersection, …/geometry/bvh_node.cuh:19-24 (recursive, both
 * children, unordered).  ref >= 0 node, ref < 0 primitive (-ref-1). */
static int tree_closest_intersection(const orc_world* w, int32_t ref, const ray_t* ray, rec_t* rec,
                                     orc_counters* cnt, int depth, int* err, rng_t* g) {
    if (ref < 0) return prim_closest_intersection(w, -ref - 1, ray, rec, cnt, g);
    if (depth > 4096) { *err = 4; return 0; }
    const orc_node* n = &w->nodes[ref];
    float d;
    if (!node_box(n, ray, rec->distance, &d, cnt)) return 0;
    int hit = tree_closest_intersection(w, n->left, ray, rec, cnt, depth + 1, err, g);
    hit |= tree_closest_intersection(w, n->right, ray, rec, cnt, depth + 1, err, g);
    return hit;
}

static inline int world_closest_intersection(const orc_world* w, const ray_t* ray, rec_t* rec, orc_counters* cnt, int* err, rng_t* g) {
    cnt->rays++;
    switch (w->kind) {
    case 0: return w->traversal == 1u ? bvh_closest_intersection_queue(w, ray, rec, cnt, err, g)
                 : w->traversal == 2u ? bvh_closest_intersection_wide4(w, ray, rec, cnt, err, g) : bvh_closest_intersection(w, ray, rec, cnt, err, g);
    case 1: return list_closest_intersection(w, ray, rec, cnt, g);
    default: return tree_closest_intersection(w, w->root, ray, rec, cnt, 0, err, g);
    }
}

/* ------------------------------------------------------------------ */
/* materials                                                           */
/* ------------------------------------------------------------------ */
/* reflectance, …/shaders/cu_materials.cuh:99-104 ; powf(x,5) -> x2,x4,x5 (see header) */
static inline float reflectance(float cos_theta, float ior_ratio) {
    float r0 = (1 - ior_ratio) / (1 + ior_ratio);
    r0 = r0 * r0;
    float x = 1 - cos_theta;
    float x2 = x * x;
    float x4 = x2 * x2;
    float x5 = x4 * x;
    return r0 + (1 - r0) * x5;
}

/* checker_texture::value, …/shaders/cu_Textures.cuh:31-39 (ivec3 truncation) */
static inline v3 checker_value(const orc_material* m, v3 pos) {
    v3 sp = muls(pos, m->param);
    int ix = (int)sp.x, iy = (int)sp.y, iz = (int)sp.z;
    int sum = 0; sum += ix; sum += iy; sum += iz;
    return (sum % 2 == 0) ? ld3(m->albedo) : ld3(m->albedo2);
}

/* perlin::noise / perlin_interp / turb, noise_texture::value, sphere::get_sphere_uv, image_texture::value of "The Next
 * Week" (extension, not in the reference): fp32, one fixed evaluation order, own sin / acos / atan2 (m_*). */
static float perlin_noise(const orc_perlin* t, v3 p) {
    float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
    float u = p.x - fx, v = p.y - fy, w = p.z - fz;
    int i = (int)fx, j = (int)fy, k = (int)fz;
    float uu = (u * u) * (3.0f - 2.0f * u), vv = (v * v) * (3.0f - 2.0f * v), ww = (w * w) * (3.0f - 2.0f * w);
    float accum = 0.0f;
    for (int di = 0; di < 2; di++)
        for (int dj = 0; dj < 2; dj++)
            for (int dk = 0; dk < 2; dk++) {
                int idx = t->perm[0][(i + di) & 255] ^ t->perm[1][(j + dj) & 255] ^ t->perm[2][(k + dk) & 255];
                const float* c = t->randvec[idx];
                float wx = u - (float)di, wy = v - (float)dj, wz = w - (float)dk;
                float d = c[0] * wx + c[1] * wy + c[2] * wz;
                float wi = di ? uu : 1.0f - uu, wj = dj ? vv : 1.0f - vv, wk = dk ? ww : 1.0f - ww;
                accum += ((wi * wj) * wk) * d;
            }
    return accum;
}
static float perlin_turb(const orc_perlin* t, v3 p, int depth) {
    float accum = 0.0f, weight = 1.0f;
    for (int i = 0; i < depth; i++) {
        accum += weight * perlin_noise(t, p);
        weight *= 0.5f;
        p = muls(p, 2.0f);
    }
    return fabsf(accum);
}
static v3 noise_value(const orc_perlin* t, v3 albedo, float scale, v3 p) {
    return muls(albedo, 1.0f + m_sinf(scale * p.z + 10.0f * perlin_turb(t, p, 7)));
}
/* image_texture::value(u, v) of "The Next Week": clamp, flip v, nearest texel, 1/255 */
static v3 image_texel(const uint8_t* image, uint32_t width, uint32_t height, float u, float v);
static v3 image_value(const uint8_t* image, uint32_t width, uint32_t height, v3 n) {
    float cy = -n.y;
    cy = cy < -1.0f ? -1.0f : (cy > 1.0f ? 1.0f : cy);
    float theta = m_acosf(cy);
    float phi = m_atan2f(-n.z, n.x) + 0x1.921fb6p+1f;
    return image_texel(image, width, height, phi / 0x1.921fb6p+2f, theta / 0x1.921fb6p+1f);
}
/* on a quad the texture coordinates are the planar coordinates of the hit, (alpha, beta) of quad::hit, recomputed from the hit point */
static v3 image_value_quad(const uint8_t* image, uint32_t width, uint32_t height, const orc_quad* q, v3 hit_p) {
    v3 planar = sub(hit_p, ld3(q->Q));
    float alpha = dot(ld3(q->w), cross(planar, ld3(q->v)));
    float beta = dot(ld3(q->w), cross(ld3(q->u), planar));
    return image_texel(image, width, height, alpha, beta);
}
static v3 image_texel(const uint8_t* image, uint32_t width, uint32_t height, float u, float v) {
    u = u < 0.0f ? 0.0f : (u > 1.0f ? 1.0f : u);
    v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
    v = 1.0f - v;
    int i = (int)(u * (float)width), j = (int)(v * (float)height);
    if (i > (int)width - 1) i = (int)width - 1;
    if (j > (int)height - 1) j = (int)height - 1;
    const uint8_t* px = image + ((size_t)j * width + (size_t)i) * 3u;
    return V((float)px[0] * 0x1.010102p-8f, (float)px[1] * 0x1.010102p-8f, (float)px[2] * 0x1.010102p-8f);
}

/* Material::Scatter for the four material classes:
 *  LambertianAbstract  …/shaders/cu_materials.cuh:52-64
 *  MetalAbstract       :77-95
 *  DielectricAbstract  :115-143
 *  LambertianTexture   :27-40 */
static int material_scatter(const orc_material* m, const ray_t* in_ray, const rec_t* rec, rng_t* g,
                            ray_t* out, v3* attenuation, const orc_world* w) {
    v3 normal = rec->normal;
    if (m->type == 4) return 0; /* diffuse_light of "The Next Week": emits (material_emitted), never scatters */
    switch (m->type) {
    case 0:
    case 3:
    case 6:   /* lambertian(noise_texture) */
    case 7: { /* lambertian(image_texture) */
        v3 ray_dir = add(normal, rng_on_unit3(g));
        if (near_zero(ray_dir)) return 0;
        out->o = ray_at(in_ray, rec->distance); out->d = ray_dir; out->time = in_ray->time;
        if (m->type == 0) *attenuation = ld3(m->albedo);
        else if (m->type == 3) *attenuation = checker_value(m, ray_at(in_ray, rec->distance));
        else if (m->type == 6) *attenuation = noise_value(w->perlin, ld3(m->albedo), m->param, ray_at(in_ray, rec->distance));
        else if (rec->prim >= 0 && (uint32_t)rec->prim >= w->n_prims)
            *attenuation = image_value_quad(w->image, w->image_width, w->image_height, &w->quads[(uint32_t)rec->prim - w->n_prims], ray_at(in_ray, rec->distance));
        else *attenuation = image_value(w->image, w->image_width, w->image_height, normal);
        return 1;
    }
    case 1: {
        v3 refl = reflect(in_ray->d, normal);
        v3 scatter_dir = add(refl, muls(rng_on_unit3(g), m->param));
        if (dot(scatter_dir, normal) < 0 || near_zero(scatter_dir)) return 0;
        out->o = ray_at(in_ray, rec->distance); out->d = scatter_dir; out->time = in_ray->time;
        *attenuation = ld3(m->albedo);
        return 1;
    }
    case 5: { /* isotropic phase function of "The Next Week" (extension): a uniformly random direction, always scatters */
        out->o = ray_at(in_ray, rec->distance); out->d = rng_on_unit3(g); out->time = in_ray->time;
        *attenuation = ld3(m->albedo);
        return 1;
    }
    default: {
        float ior = m->param;
        int hit_backface = dot(in_ray->d, normal) > 0; /* isBackfacing, ray_data.cuh:44-46 */
        if (hit_backface) normal = neg(normal);
        float ior_ratio = hit_backface ? ior : 1 / ior;
        v3 unit_dir = normalize(in_ray->d);
        float cos_theta = fminf(dot(neg(unit_dir), normal), 1.0f);
        float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
        float reflect_prob = reflectance(cos_theta, ior_ratio);
        v3 scatter_dir;
        if (ior_ratio * sin_theta > 1.0f || reflect_prob > rng_next(g))
            scatter_dir = reflect(unit_dir, normal);
        else
            scatter_dir = refract(unit_dir, normal, ior_ratio);
        out->o = ray_at(in_ray, rec->distance); out->d = scatter_dir; out->time = in_ray->time;
        *attenuation = ld3(m->albedo);
        return 1;
    }
    }
}

/* ------------------------------------------------------------------ */
/* cameras, …/shaders/cu_Cameras.cuh                                   */
/* ------------------------------------------------------------------ */
static void camera_basis(const float lookfrom[3], const float lookat[3], const float up[3], float vfov,
                         float aspect, orc_camera* c, int prescale) {
    float theta = radians(vfov);
    float vh = tanf(theta * 0.5f);
    float vw = vh * aspect;
    v3 w = normalize(sub(ld3(lookat), ld3(lookfrom)));
    v3 u = normalize(cross(ld3(up), w));
    if (prescale) u = muls(u, vw);
    v3 v = normalize(cross(w, u));
    if (prescale) v = muls(v, vh);
    memset(c, 0, sizeof(*c));
    st3(c->o, ld3(lookfrom)); st3(c->u, u); st3(c->v, v); st3(c->w, w);
    c->viewport_width = vw; c->viewport_height = vh;
    c->t0 = 0.0f; c->t1 = 1.0f;
}
/* PinholeCamera ctor :16-25 */
void orc_camera_pinhole(const float lookfrom[3], const float lookat[3], const float up[3], float vfov,
                        float aspect, orc_camera* out) {
    camera_basis(lookfrom, lookat, up, vfov, aspect, out, 1);
    out->type = 0;
}
/* DefocusBlurCamera ctor :40-52 */
void orc_camera_defocus(const float lookfrom[3], const float lookat[3], const float up[3], float vfov,
                        float aspect, float aperture, float focus_dist, orc_camera* out) {
    camera_basis(lookfrom, lookat, up, vfov, aspect, out, 0);
    out->type = 1;
    out->lens_radius = aperture * 0.5f;
    out->focus_dist = focus_dist;
}
/* MotionBlurCamera ctor :73-85 */
void orc_camera_motion(const float lookfrom[3], const float lookat[3], const float up[3], float vfov,
                       float aspect, float t0, float t1, orc_camera* out) {
    camera_basis(lookfrom, lookat, up, vfov, aspect, out, 1);
    out->type = 2;
    out->t0 = t0; out->t1 = t1;
}

/* sample_ray :27-30 (pinhole), :54-64 (defocus), :87-89 (motion) */
static inline ray_t camera_sample_ray(const orc_camera* c, float s, float t, rng_t* g) {
    ray_t r;
    v3 o = ld3(c->o), u = ld3(c->u), v = ld3(c->v), w = ld3(c->w);
    if (c->type == 1) {
        float dx, dy;
        rng_in_unit2(g, &dx, &dy);
        v3 offset = add(muls(u, dx), muls(v, dy));
        offset = muls(offset, c->lens_radius);
        v3 forward = muls(w, c->focus_dist);
        v3 hori = muls(muls(u, c->viewport_width), c->focus_dist);
        v3 vert = muls(muls(v, c->viewport_height), c->focus_dist);
        r.o = add(o, offset);
        r.d = sub(add(add(forward, muls(hori, s)), muls(vert, t)), offset);
        r.time = 0.0f;
    } else {
        r.o = o;
        r.d = add(add(w, muls(u, s)), muls(v, t));
        r.time = (c->type == 2) ? mix1(c->t0, c->t1, rng_next(g)) : 0.0f;
    }
    return r;
}

/* ------------------------------------------------------------------ */
/* sample_world, main/src/Renderer.cu:139-181                          */
/* ------------------------------------------------------------------ */
/* The emission / background hooks are the reference's own commented placeholders (Renderer.cu:142,152,157,163,179:
 * `accum_radiance`); with no emissive material and background 0 this is exactly the live sample_world. */
static v3 sample_world(const orc_world* w, ray_t cur_ray, uint32_t max_depth, rng_t* g, orc_counters* cnt, int* err) {
    v3 accum_attenuation = V(1.0f, 1.0f, 1.0f);
    v3 accum_radiance = V(0.0f, 0.0f, 0.0f);
    for (uint32_t i = 0; i < max_depth; i++) {
        rec_t rec; rec.distance = ORC_MISS_DIST; rec.prim = -1; rec.mat = 0; rec.normal = V(0, 0, 0);
        if (!world_closest_intersection(w, &cur_ray, &rec, cnt, err, g)) {
            v3 sky;
            if (w->background == 1) sky = ld3(w->background_color);
            else {
                float t = normalize(cur_ray.d).y * 0.5f + 0.5f;
                sky = lerp3(V(0.1f, 0.2f, 0.4f), V(0.9f, 0.9f, 0.99f), t);
            }
            return add(mul(accum_attenuation, sky), accum_radiance);
        }
        cnt->shaded_hits++;
        const orc_material* m = &w->materials[rec.mat];
        if (m->type == 4) accum_radiance = add(accum_radiance, mul(accum_attenuation, ld3(m->albedo)));
        ray_t scattered; v3 attenuation;
        if (!material_scatter(m, &cur_ray, &rec, g, &scattered, &attenuation, w))
            return accum_radiance;
        accum_attenuation = mul(accum_attenuation, attenuation);
        cur_ray = scattered;
        cur_ray.o = add(cur_ray.o, muls(cur_ray.d, 0.001f));
    }
    return accum_radiance;
}

/* one sample of render_kernel's loop body, Renderer.cu:198-204.  The RNG is
 * keyed per (pixel, sample) instead of one XORWOW stream per pixel
 * (Renderer.cu:191) — SURVEY.md Appendix A item 7. */
static inline v3 one_sample(const orc_world* w, const orc_camera* cam, uint32_t width, uint32_t height,
                            uint32_t max_depth, uint64_t seed, uint32_t gid, uint32_t s, orc_counters* cnt, int* err) {
    uint32_t x = gid % width, y = gid / width;
    float psx = 1.0f / (float)width, psy = 1.0f / (float)height;
    float ndcx = ((float)x + 0.5f) * psx * 2.0f - 1.0f;
    float ndcy = ((float)y + 0.5f) * psy * 2.0f - 1.0f;
    rng_t g; rng_init(&g, seed, gid, s, 0u);
    float jx, jy;
    rng_in_unit2(&g, &jx, &jy);
    float sx = ndcx + jx * psx;
    float sy = ndcy + jy * psy;
    ray_t ray = camera_sample_ray(cam, sx, sy, &g);
    v3 rad = sample_world(w, ray, max_depth, &g, cnt, err);
    cnt->samples++;
    cnt->rng_draws += g.draws;
    return rad;
}

/* render_kernel, Renderer.cu:183-217 (one pixel) */
static void render_pixel(const orc_world* w, const orc_camera* cam, uint32_t width, uint32_t height,
                         uint32_t spp, uint32_t max_depth, uint64_t seed, uint32_t x, uint32_t y,
                         float* out_rgba, orc_counters* cnt, int* err) {
    uint32_t gid = y * width + x;
    v3 radiance = V(0.0f, 0.0f, 0.0f);
    for (uint32_t s = 0; s < spp; s++)
        radiance = add(radiance, one_sample(w, cam, width, height, max_depth, seed, gid, s, cnt, err));
    radiance = muls(radiance, 1.0f / (float)spp);
    v3 col = clamp01_sqrt(radiance);
    float* o = out_rgba + (size_t)gid * 4;
    o[0] = col.x; o[1] = col.y; o[2] = col.z; o[3] = 1.0f;
}

typedef struct {
    const orc_world* w; const orc_camera* cam;
    uint32_t width, height, spp, max_depth; uint64_t seed;
    float* out; uint32_t* next_row; orc_counters cnt; int err;
} job_t;

static void* render_worker(void* arg) {
    job_t* j = (job_t*)arg;
    for (;;) {
        uint32_t y = __atomic_fetch_add(j->next_row, 1u, __ATOMIC_RELAXED);
        if (y >= j->height) break;
        for (uint32_t x = 0; x < j->width; x++)
            render_pixel(j->w, j->cam, j->width, j->height, j->spp, j->max_depth, j->seed, x, y, j->out, &j->cnt, &j->err);
    }
    return NULL;
}

int orc_render(const orc_world* w, const orc_camera* cam, uint32_t width, uint32_t height, uint32_t spp,
               uint32_t max_depth, uint64_t seed, int n_threads, float* out_rgba, orc_counters* counters) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 1024) n_threads = 1024;
    job_t* jobs = (job_t*)calloc((size_t)n_threads, sizeof(job_t));
    pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(pthread_t));
    uint32_t next_row = 0;
    for (int i = 0; i < n_threads; i++) {
        jobs[i].w = w; jobs[i].cam = cam; jobs[i].width = width; jobs[i].height = height;
        jobs[i].spp = spp; jobs[i].max_depth = max_depth; jobs[i].seed = seed;
        jobs[i].out = out_rgba; jobs[i].next_row = &next_row;
    }
    if (n_threads == 1) render_worker(&jobs[0]);
    else {
        /* ORC_PIN_THREADS=1 (bench.py's cpu_baseline leg): worker i is pinned to the i-th CPU this process may run on, so the timed
         * baseline uses n_threads distinct cores (on the EPYC hosts of the GPU boxes the first CPUs are distinct physical cores of socket 0) */
        const char* pin = getenv("ORC_PIN_THREADS");
        cpu_set_t allowed;
        int have_mask = pin && pin[0] == '1' && sched_getaffinity(0, sizeof(allowed), &allowed) == 0;
        int cpu = -1;
        for (int i = 0; i < n_threads; i++) {
            pthread_create(&th[i], NULL, render_worker, &jobs[i]);
            if (have_mask) {
                do { cpu++; } while (cpu < CPU_SETSIZE && !CPU_ISSET(cpu, &allowed));
                if (cpu < CPU_SETSIZE) { cpu_set_t one; CPU_ZERO(&one); CPU_SET(cpu, &one); (void)pthread_setaffinity_np(th[i], sizeof(one), &one); }
            }
        }
        for (int i = 0; i < n_threads; i++) pthread_join(th[i], NULL);
    }
    int err = 0;
    orc_counters total; memset(&total, 0, sizeof(total));
    for (int i = 0; i < n_threads; i++) {
        if (jobs[i].err) err = jobs[i].err;
        total.samples += jobs[i].cnt.samples; total.rays += jobs[i].cnt.rays;
        total.box_tests += jobs[i].cnt.box_tests; total.leaf_tests += jobs[i].cnt.leaf_tests;
        total.shaded_hits += jobs[i].cnt.shaded_hits; total.rng_draws += jobs[i].cnt.rng_draws;
        if (jobs[i].cnt.max_stack > total.max_stack) total.max_stack = jobs[i].cnt.max_stack;
    }
    if (counters) *counters = total;
    free(jobs); free(th);
    return err;
}

/* ------------------------------------------------------------------ */
/* batch entry points (twins of rt_probe_*)                            */
/* ------------------------------------------------------------------ */
static inline ray_t ld_ray6(const float* p) { ray_t r; r.o = ld3(p); r.d = ld3(p + 3); r.time = 0.0f; return r; }
static inline ray_t ld_ray7(const float* p) { ray_t r; r.o = ld3(p); r.d = ld3(p + 3); r.time = p[6]; return r; }
static inline void st_ray7(float* p, const ray_t* r) { st3(p, r->o); st3(p + 3, r->d); p[6] = r->time; }

void orc_aabb_batch(size_t n, const float* boxes, const float* rays, const float* max_dist, int32_t* out_hit, float* out_dist) {
    for (size_t i = 0; i < n; i++) {
        ray_t r = ld_ray6(rays + 6 * i);
        float d = 0.0f;
        out_hit[i] = aabb_intersects(ld3(boxes + 6 * i), ld3(boxes + 6 * i + 3), &r, max_dist[i], &d);
        out_dist[i] = d;
    }
}
void orc_sphere_batch(size_t n, const float* rays, const float* spheres, float* out_t) {
    for (size_t i = 0; i < n; i++) {
        ray_t r = ld_ray6(rays + 6 * i);
        out_t[i] = sphere_closest_intersection(&r, ld3(spheres + 4 * i), spheres[4 * i + 3]);
    }
}
int orc_trace_batch(const orc_world* w, size_t n, const float* rays, int32_t* out_hit, float* out_t, int32_t* out_prim, float* out_normal) {
    orc_counters cnt; memset(&cnt, 0, sizeof(cnt));
    int err = 0;
    for (size_t i = 0; i < n; i++) {
        ray_t r = ld_ray7(rays + 7 * i);
        rec_t rec; rec.distance = ORC_MISS_DIST; rec.prim = -1; rec.mat = 0; rec.normal = V(0, 0, 0);
        rng_t g; rng_init(&g, 0u, (uint32_t)i, 0u, 0x7ACEu); /* only a constant medium draws from it */
        out_hit[i] = world_closest_intersection(w, &r, &rec, &cnt, &err, &g);
        out_t[i] = rec.distance; out_prim[i] = rec.prim; st3(out_normal + 3 * i, rec.normal);
    }
    return err;
}
void orc_scatter_batch(uint64_t seed, size_t n, const orc_material* mats, const float* rays, const float* dist,
                       const float* normals, const uint32_t* keys, int32_t* out_scattered, float* out_rays,
                       float* out_atten, uint32_t* out_draws) {
    for (size_t i = 0; i < n; i++) {
        ray_t in = ld_ray7(rays + 7 * i);
        rec_t rec; rec.distance = dist[i]; rec.normal = ld3(normals + 3 * i); rec.prim = 0; rec.mat = 0;
        rng_t g; rng_init(&g, seed, keys[2 * i], keys[2 * i + 1], 0u);
        ray_t out; out.o = V(0, 0, 0); out.d = V(0, 0, 0); out.time = 0.0f;
        v3 att = V(0, 0, 0);
        out_scattered[i] = material_scatter(&mats[i], &in, &rec, &g, &out, &att, NULL);
        st_ray7(out_rays + 7 * i, &out); st3(out_atten + 3 * i, att); out_draws[i] = g.draws;
    }
}
void orc_camera_batch(uint64_t seed, const orc_camera* cam, size_t n, const float* st, const uint32_t* keys,
                      float* out_rays, uint32_t* out_draws) {
    for (size_t i = 0; i < n; i++) {
        rng_t g; rng_init(&g, seed, keys[2 * i], keys[2 * i + 1], 0u);
        ray_t r = camera_sample_ray(cam, st[2 * i], st[2 * i + 1], &g);
        st_ray7(out_rays + 7 * i, &r); out_draws[i] = g.draws;
    }
}
int orc_radiance_batch(const orc_world* w, const orc_camera* cam, uint32_t width, uint32_t height, uint32_t max_depth,
                       uint64_t seed, size_t n, const uint32_t* keys, float* out_radiance) {
    orc_counters cnt; memset(&cnt, 0, sizeof(cnt));
    int err = 0;
    for (size_t i = 0; i < n; i++)
        st3(out_radiance + 3 * i, one_sample(w, cam, width, height, max_depth, seed, keys[2 * i], keys[2 * i + 1], &cnt, &err));
    return err;
}
/* render_kernel for a list of pixels (full spp loop, mean, clamp, gamma): lets a test check a sparse
 * sample of a full-size frame without rendering all of it on the CPU. */
int orc_render_pixels(const orc_world* w, const orc_camera* cam, uint32_t width, uint32_t height, uint32_t spp,
                      uint32_t max_depth, uint64_t seed, size_t n, const uint32_t* gids, float* out_rgba) {
    orc_counters cnt; memset(&cnt, 0, sizeof(cnt));
    int err = 0;
    float* tmp = (float*)malloc((size_t)width * height * 4 * sizeof(float));
    for (size_t i = 0; i < n; i++) {
        uint32_t gid = gids[i];
        render_pixel(w, cam, width, height, spp, max_depth, seed, gid % width, gid / width, tmp, &cnt, &err);
        memcpy(out_rgba + 4 * i, tmp + (size_t)gid * 4, 4 * sizeof(float));
    }
    free(tmp);
    return err;
}
/* SphereTest::_pixel_ground_truth, google_testing/test.cpp:87-106 */
void orc_sphere_index(const orc_camera* cam, uint32_t width, uint32_t height, size_t n_spheres, const float* spheres, int32_t* out_index) {
    for (uint32_t y = 0; y < height; y++)
        for (uint32_t x = 0; x < width; x++) {
            float u = (float)x / ((float)width - 1.0f) * 2 - 1;
            float v = (float)y / ((float)height - 1.0f) * 2 - 1;
            /* PinholeCamera::sample_ray, cu_Cameras.cuh:27-30 */
            ray_t ray;
            ray.o = ld3(cam->o);
            ray.d = add(add(ld3(cam->w), muls(ld3(cam->u), u)), muls(ld3(cam->v), v));
            ray.time = 0.0f;
            float best = ORC_MISS_DIST;
            int32_t result = -1;
            for (size_t i = 0; i < n_spheres; i++) {
                float dist = sphere_closest_intersection(&ray, ld3(spheres + 4 * i), spheres[4 * i + 3]);
                if (dist < best) { result = (int32_t)i; best = dist; }
            }
            out_index[(size_t)y * width + x] = result;
        }
}

/* ------------------------------------------------------------------ */
/* scene generation and BVH builders (host side of the path)           */
/* ------------------------------------------------------------------ */
struct orc_scene {
    orc_prim* prims; size_t n_prims;
    orc_quad* quads; size_t n_quads;
    uint32_t background; float background_color[3];
    orc_material* mats; size_t n_mats;
    orc_node* nodes; size_t n_nodes, cap_nodes;
    orc_perlin* perlin;
    uint8_t* image; uint32_t image_w, image_h;
    orc_world world;
};

typedef struct { v3 mn, mx; } box_t;
static inline box_t box_empty(void) { box_t b = {V(1e9f, 1e9f, 1e9f), V(-1e9f, -1e9f, -1e9f)}; return b; } /* aabb(), aabb.cuh:17 */
static inline box_t box_union(box_t a, box_t b) { box_t r = {vmin(a.mn, b.mn), vmax(a.mx, b.mx)}; return r; } /* aabb.cuh:19,24 */
/* getSphereBounds / getMovingSphereBounds, SphereHittable.cu:52-54,85-89 */
static box_t prim_bounds(const orc_prim* p) {
    v3 r = V(p->radius, p->radius, p->radius);
    box_t b0 = {sub(ld3(p->c0), r), add(ld3(p->c0), r)};
    if (!(p->mat & ORC_PRIM_MOVING)) return b0;
    box_t b1 = {sub(ld3(p->c1), r), add(ld3(p->c1), r)};
    return box_union(b0, b1);
}
/* aabb::longest_axis, aabb.cuh:46-53 */
static int box_longest_axis(box_t b) {
    v3 s = sub(b.mx, b.mn);
    s = V(fabsf(s.x), fabsf(s.y), fabsf(s.z));
    if (s.x > s.y) return s.x > s.z ? 0 : 2;
    return s.y > s.z ? 1 : 2;
}
/* aabb::surface_area, aabb.cuh:55-64 */
static float box_surface_area(box_t b) {
    v3 s = sub(b.mx, b.mn);
    if (s.x < 0 || s.y < 0 || s.z < 0) return 0.0f;
    float cost = 0.0f;
    cost += s.x * s.y; cost += s.x * s.z; cost += s.y * s.z;
    return 2.0f * cost;
}
static inline v3 box_centroid(box_t b) { return muls(add(b.mx, b.mn), 0.5f); } /* aabb.cuh:66-68 */
static inline float v3_axis(v3 a, int ax) { return ax == 0 ? a.x : (ax == 1 ? a.y : a.z); }

/* quad cached quantities and bounds, "The Next Week" quad(Q,u,v): n = cross(u,v), normal = unit(n), D = dot(normal,Q),
 * w = n / dot(n,n); bbox = box(Q, Q+u+v) U box(Q+u, Q+v), every axis padded to at least 0.0001 */
static void quad_finalize(orc_quad* q) {
    v3 n = cross(ld3(q->u), ld3(q->v));
    v3 normal = normalize(n);
    st3(q->normal, normal);
    q->D = dot(normal, ld3(q->Q));
    st3(q->w, divs(n, dot(n, n)));
    q->pad0 = q->pad1 = q->pad2 = 0.0f;
}
static box_t box_of_points(v3 a, v3 b) { box_t r = {vmin(a, b), vmax(a, b)}; return r; }
static box_t quad_bounds(const orc_quad* q) {
    v3 Q = ld3(q->Q), u = ld3(q->u), v = ld3(q->v);
    box_t b = box_union(box_of_points(Q, add(add(Q, u), v)), box_of_points(add(Q, u), add(Q, v)));
    const float delta = 0.0001f;
    if (b.mx.x - b.mn.x < delta) { b.mn.x -= delta / 2; b.mx.x += delta / 2; }
    if (b.mx.y - b.mn.y < delta) { b.mn.y -= delta / 2; b.mx.y += delta / 2; }
    if (b.mx.z - b.mn.z < delta) { b.mn.z -= delta / 2; b.mx.z += delta / 2; }
    return b;
}

typedef struct { box_t b; int is_quad; orc_prim p; orc_quad q; } item_t;
typedef struct {
    item_t* arr; item_t* tmp;
    orc_scene* s;
} builder_t;

static int32_t push_node(orc_scene* s, box_t b, int32_t left, int32_t right) {
    if (s->n_nodes == s->cap_nodes) {
        s->cap_nodes = s->cap_nodes ? s->cap_nodes * 2 : 64;
        s->nodes = (orc_node*)realloc(s->nodes, s->cap_nodes * sizeof(orc_node));
    }
    orc_node* n = &s->nodes[s->n_nodes];
    st3(n->min, b.mn); st3(n->max, b.mx); n->left = left; n->right = right;
    return (int32_t)(s->n_nodes++);
}
/* _get_partition_bounds, BVH.cu:306-312 */
static box_t partition_bounds(builder_t* B, int start, int end) {
    box_t b = box_empty();
    for (int i = start; i < end; i++) b = box_union(b, B->arr[i].b);
    return b;
}
/* std::sort by bounds.min[axis] (BVH.cu:195-199, aabb.cuh:78-88).  std::sort is
 * unstable; the build fixes the tie order by using a STABLE merge sort.     */
static void stable_sort_axis(builder_t* B, int start, int end, int axis) {
    int n = end - start;
    if (n < 2) return;
    item_t* a = B->arr + start; item_t* t = B->tmp + start;
    for (int width = 1; width < n; width *= 2) {
        for (int lo = 0; lo < n; lo += 2 * width) {
            int mid = lo + width < n ? lo + width : n;
            int hi = lo + 2 * width < n ? lo + 2 * width : n;
            int i = lo, j = mid, k = lo;
            while (i < mid && j < hi) {
                /* take from the right run only if strictly less (stability) */
                if (v3_axis(a[j].b.mn, axis) < v3_axis(a[i].b.mn, axis)) t[k++] = a[j++];
                else t[k++] = a[i++];
            }
            while (i < mid) t[k++] = a[i++];
            while (j < hi) t[k++] = a[j++];
        }
        memcpy(a, t, (size_t)n * sizeof(item_t));
    }
}
/* _build_bvh_rec1, BVH.cu:180-210 */
static int32_t build_rec1(builder_t* B, int start, int end) {
    box_t bounds = partition_bounds(B, start, end);
    int axis = box_longest_axis(bounds);
    if (end - start == 1) return push_node(B->s, bounds, -1, start);
    stable_sort_axis(B, start, end, axis);
    int mid = (start + end) / 2;
    int32_t l = build_rec1(B, start, mid);
    int32_t r = build_rec1(B, mid, end);
    return push_node(B->s, bounds, l, r);
}
/* _find_optimal_split, BVH.cu:241-279 */
static void find_optimal_split(builder_t* B, int start, int end, box_t bounds, int* best_axis, float* best_split) {
    const int split_points = 16;
    float best_cost = 3.402823466e+38F;
    *best_axis = 0; *best_split = 0.0f;
    for (int axis = 0; axis < 3; axis++)
        for (int split = 0; split < split_points; split++) {
            float pos = ((float)split + 1.0f) / ((float)split_points + 1.0f);
            pos = mix1(v3_axis(bounds.mn, axis), v3_axis(bounds.mx, axis), pos);
            box_t lb = box_empty(), rb = box_empty();
            int lc = 0, rc = 0;
            for (int i = start; i < end; i++) {
                box_t b = B->arr[i].b;
                if (v3_axis(box_centroid(b), axis) < pos) { lb = box_union(lb, b); lc++; }
                else { rb = box_union(rb, b); rc++; }
            }
            float cost = box_surface_area(lb) * (float)lc + box_surface_area(rb) * (float)rc;
            if (cost < best_cost) { best_cost = cost; *best_axis = axis; *best_split = pos; }
        }
}
/* _partition_by_split, BVH.cu:281-304 */
static int partition_by_split(builder_t* B, int start, int end, int axis, float split_pos) {
    int i = start, j = end;
    while (i < j) {
        if (v3_axis(box_centroid(B->arr[i].b), axis) < split_pos) i++;
        else { item_t t = B->arr[i]; B->arr[i] = B->arr[--j]; B->arr[j] = t; }
    }
    return i;
}
/* _build_bvh_rec2, BVH.cu:212-239.  The reference recurses forever when a
 * split leaves one side empty; the build falls back to the median split of
 * rec1 for that range (documented fix).                          */
static int32_t build_rec2(builder_t* B, int start, int end) {
    box_t bounds = partition_bounds(B, start, end);
    if (end - start == 1) return push_node(B->s, bounds, -1, start);
    int axis; float split;
    find_optimal_split(B, start, end, bounds, &axis, &split);
    int mid = partition_by_split(B, start, end, axis, split);
    if (mid == start || mid == end) {
        stable_sort_axis(B, start, end, box_longest_axis(bounds));
        mid = (start + end) / 2;
    }
    int32_t l = build_rec2(B, start, mid);
    int32_t r = build_rec2(B, mid, end);
    return push_node(B->s, bounds, l, r);
}
/* BuildBVH_BottomUp + _find_optimal_merge + _merge_nodes, BVH.cu:315-384 */
typedef struct { box_t b; int count; int32_t idx; } bnode_t;
static int32_t build_bottom_up(builder_t* B, int n) {
    bnode_t* bn = (bnode_t*)malloc((size_t)n * sizeof(bnode_t));
    int m = n;
    for (int i = 0; i < n; i++) {
        bn[i].b = B->arr[i].b; bn[i].count = 1;
        bn[i].idx = push_node(B->s, B->arr[i].b, -1, i);
    }
    while (m > 1) {
        float best_cost = 3.402823466e+38F;
        int ba = 0, bb = 1;
        for (int a = 0; a < m; a++)
            for (int b = a + 1; b < m; b++) {
                box_t nb = box_union(bn[a].b, bn[b].b);
                float cost = box_surface_area(nb) * (float)(bn[a].count + bn[b].count);
                if (cost < best_cost) { ba = a; bb = b; best_cost = cost; }
            }
        bnode_t merged;
        merged.b = box_union(bn[ba].b, bn[bb].b);
        merged.count = bn[ba].count + bn[bb].count;
        merged.idx = push_node(B->s, merged.b, bn[ba].idx, bn[bb].idx);
        /* erase bb then ba (ba < bb), push_back merged */
        memmove(&bn[bb], &bn[bb + 1], (size_t)(m - bb - 1) * sizeof(bnode_t)); m--;
        memmove(&bn[ba], &bn[ba + 1], (size_t)(m - ba - 1) * sizeof(bnode_t)); m--;
        bn[m++] = merged;
    }
    int32_t root = bn[0].idx;
    free(bn);
    return root;
}

static uint32_t tree_leaf_depth(const orc_node* nodes, int32_t idx) {
    if (nodes[idx].left == -1) return 0;
    uint32_t a = tree_leaf_depth(nodes, nodes[idx].left), b = tree_leaf_depth(nodes, nodes[idx].right);
    return 1 + (a > b ? a : b);
}

static void finish_scene(orc_scene* s, int builder) {
    size_t ns = s->n_prims, nq = s->n_quads, n = ns + nq;
    orc_world* w = &s->world;
    memset(w, 0, sizeof(*w));
    box_t wb = box_empty();
    for (size_t i = 0; i < ns; i++) wb = box_union(wb, prim_bounds(&s->prims[i]));
    for (size_t i = 0; i < nq; i++) wb = box_union(wb, quad_bounds(&s->quads[i]));
    if (builder == 3) {
        w->kind = 1; w->root = 0;
    } else {
        builder_t B; B.s = s;
        B.arr = (item_t*)malloc(n * sizeof(item_t)); B.tmp = (item_t*)malloc(n * sizeof(item_t));
        for (size_t i = 0; i < ns; i++) { B.arr[i].is_quad = 0; B.arr[i].p = s->prims[i]; B.arr[i].b = prim_bounds(&s->prims[i]); }
        for (size_t i = 0; i < nq; i++) { B.arr[ns + i].is_quad = 1; B.arr[ns + i].q = s->quads[i]; B.arr[ns + i].b = quad_bounds(&s->quads[i]); }
        int32_t root;
        if (builder == 0) root = build_rec1(&B, 0, (int)n);
        else if (builder == 1) root = build_rec2(&B, 0, (int)n);
        else root = build_bottom_up(&B, (int)n);
        /* hittables = sorted order (BVH.cu:174-177), kept per kind: spheres first, then quads; a leaf's index is remapped */
        int32_t* unified = (int32_t*)malloc(n * sizeof(int32_t));
        size_t si = 0, qi = 0;
        for (size_t i = 0; i < n; i++) {
            if (B.arr[i].is_quad) { s->quads[qi] = B.arr[i].q; unified[i] = (int32_t)(ns + qi); qi++; }
            else { s->prims[si] = B.arr[i].p; unified[i] = (int32_t)si; si++; }
        }
        for (size_t i = 0; i < s->n_nodes; i++)
            if (s->nodes[i].left == -1) s->nodes[i].right = unified[s->nodes[i].right];
        free(unified); free(B.arr); free(B.tmp);
        w->kind = 0; w->root = root;
        w->max_stack = tree_leaf_depth(s->nodes, root) + 1;
        wb.mn = ld3(s->nodes[root].min); wb.mx = ld3(s->nodes[root].max);
    }
    w->n_nodes = (uint32_t)s->n_nodes; w->n_prims = (uint32_t)ns; w->n_materials = (uint32_t)s->n_mats;
    st3(w->bounds_min, wb.mn); st3(w->bounds_max, wb.mx);
    w->nodes = s->nodes; w->prims = s->prims; w->materials = s->mats;
    w->quads = s->quads; w->n_quads = (uint32_t)nq;
    w->background = s->background; st3(w->background_color, ld3(s->background_color));
}

orc_scene* orc_scene_from_arrays_ext(size_t n_prims, const orc_prim* prims, size_t n_quads, const orc_quad* quads, size_t n_mats,
                                     const orc_material* mats, int builder, uint32_t background, const float background_color[3]) {
    orc_scene* s = (orc_scene*)calloc(1, sizeof(orc_scene));
    s->prims = (orc_prim*)malloc((n_prims + 1) * sizeof(orc_prim)); if (n_prims) memcpy(s->prims, prims, n_prims * sizeof(orc_prim));
    s->n_prims = n_prims;
    s->quads = (orc_quad*)malloc((n_quads + 1) * sizeof(orc_quad)); if (n_quads) memcpy(s->quads, quads, n_quads * sizeof(orc_quad));
    s->n_quads = n_quads;
    for (size_t i = 0; i < n_quads; i++) quad_finalize(&s->quads[i]);
    s->mats = (orc_material*)malloc(n_mats * sizeof(orc_material)); memcpy(s->mats, mats, n_mats * sizeof(orc_material)); s->n_mats = n_mats;
    s->background = background;
    if (background_color) st3(s->background_color, ld3(background_color));
    finish_scene(s, builder);
    return s;
}
orc_scene* orc_scene_from_arrays(size_t n_prims, const orc_prim* prims, size_t n_mats, const orc_material* mats, int builder) {
    return orc_scene_from_arrays_ext(n_prims, prims, 0, NULL, n_mats, mats, builder, 0, NULL);
}

static void scene_add(orc_prim* prims, orc_material* mats, size_t* n, v3 c0, v3 c1, float r, int moving,
                      uint32_t type, v3 albedo, float param) {
    orc_prim* p = &prims[*n]; orc_material* m = &mats[*n];
    st3(p->c0, c0); st3(p->c1, c1); p->radius = r; p->mat = (uint32_t)(*n) | (moving ? ORC_PRIM_MOVING : 0u);
    st3(m->albedo, albedo); m->param = param; st3(m->albedo2, V(0, 0, 0)); m->type = type;
    (*n)++;
}

/* SceneBook2BVH::Factory::_populate_world, Scenes.cu:219-270 (moving = 1), and
 * the disabled SceneBook1 variant, Scenes.cu:57-115, with static Lambertians
 * (moving = 0; the centre1 draw is still consumed so both scenes share one
 * layout, as google_testing/test.cpp:41-51 does).  The uniforms come from the
 * build's host stream (stream id 0x5CE9E5) instead of cuHostRND.            */
static orc_scene* book_scene(uint64_t seed, int moving) {
    orc_prim* prims = (orc_prim*)calloc(488, sizeof(orc_prim));
    orc_material* mats = (orc_material*)calloc(488, sizeof(orc_material));
    size_t n = 0;
    rng_t g; rng_init(&g, seed, 0u, 0u, 0x5CE9E5u);
#define RND rng_next(&g)
    scene_add(prims, mats, &n, V(0, -1000, 0), V(0, -1000, 0), 1000.0f, 0, 0, V(0.5f, 0.5f, 0.5f), 0.0f);
    for (int a = -11; a < 11; a++)
        for (int b = -11; b < 11; b++) {
            float choose_mat = RND;
            float cx = (float)a + RND;
            float cz = (float)b + RND;
            v3 center = V(cx, 0.2f, cz);
            if (choose_mat < 0.8f) {
                float r0 = RND, r1 = RND, r2 = RND, r3 = RND, r4 = RND, r5 = RND;
                v3 albedo = V(r0 * r1, r2 * r3, r4 * r5);
                float rc = RND;
                v3 center1 = add(center, V(0, rc * 0.5f, 0));
                scene_add(prims, mats, &n, center, moving ? center1 : center, 0.2f, moving, 0, albedo, 0.0f);
            } else if (choose_mat < 0.95f) {
                float r0 = RND, r1 = RND, r2 = RND, r3 = RND;
                v3 albedo = V(0.5f * (1.0f + r0), 0.5f * (1.0f + r1), 0.5f * (1.0f + r2));
                scene_add(prims, mats, &n, center, center, 0.2f, 0, 1, albedo, 0.5f * r3);
            } else {
                scene_add(prims, mats, &n, center, center, 0.2f, 0, 2, V(1.0f, 1.0f, 1.0f), 1.5f);
            }
        }
#undef RND
    scene_add(prims, mats, &n, V(0, 1, 0), V(0, 1, 0), 1.0f, 0, 2, V(1.0f, 1.0f, 1.0f), 1.5f);
    scene_add(prims, mats, &n, V(-4, 1, 0), V(-4, 1, 0), 1.0f, 0, 0, V(0.4f, 0.2f, 0.1f), 0.0f);
    scene_add(prims, mats, &n, V(4, 1, 0), V(4, 1, 0), 1.0f, 0, 1, V(0.7f, 0.6f, 0.5f), 0.0f);
    orc_scene* s = orc_scene_from_arrays(n, prims, n, mats, 0);
    free(prims); free(mats);
    return s;
}
orc_scene* orc_scene_book1_final(uint64_t seed) { return book_scene(seed, 0); }
orc_scene* orc_scene_book2_moving(uint64_t seed) { return book_scene(seed, 1); }

/* Book-1 three-spheres scene (config 1; not in the reference's source, layout
 * from SURVEY.md §8d) as a HittableList.                                      */
orc_scene* orc_scene_three_spheres(void) {
    orc_prim prims[5]; orc_material mats[5]; size_t n = 0;
    memset(prims, 0, sizeof(prims)); memset(mats, 0, sizeof(mats));
    scene_add(prims, mats, &n, V(0, -100.5f, -1), V(0, -100.5f, -1), 100.0f, 0, 0, V(0.8f, 0.8f, 0.0f), 0.0f);
    scene_add(prims, mats, &n, V(0, 0, -1.2f), V(0, 0, -1.2f), 0.5f, 0, 0, V(0.1f, 0.2f, 0.5f), 0.0f);
    scene_add(prims, mats, &n, V(-1, 0, -1), V(-1, 0, -1), 0.5f, 0, 2, V(1.0f, 1.0f, 1.0f), 1.5f);
    scene_add(prims, mats, &n, V(-1, 0, -1), V(-1, 0, -1), 0.4f, 0, 2, V(1.0f, 1.0f, 1.0f), 1.0f / 1.5f);
    scene_add(prims, mats, &n, V(1, 0, -1), V(1, 0, -1), 0.5f, 0, 1, V(0.8f, 0.6f, 0.2f), 1.0f);
    return orc_scene_from_arrays(n, prims, n, mats, 3);
}
/* Cornell box of "Ray Tracing: The Next Week" (BASELINE.json configs[3]); the two boxes are built as 6 quads each and
 * rotated about y / translated on the host (the book wraps them in rotate_y / translate instances instead). */
static void cornell_quad(orc_quad* quads, size_t* n, v3 Q, v3 u, v3 v, uint32_t mat) {
    orc_quad* q = &quads[(*n)++];
    memset(q, 0, sizeof(*q));
    st3(q->Q, Q); st3(q->u, u); st3(q->v, v); q->mat = mat;
}
static v3 rot_y(v3 p, float c, float s) { return V(c * p.x + s * p.z, p.y, -s * p.x + c * p.z); }
static void cornell_box(orc_quad* quads, size_t* n, v3 a, v3 b, float degrees, v3 offset, uint32_t mat) {
    v3 mn = vmin(a, b), mx = vmax(a, b);
    v3 dx = V(mx.x - mn.x, 0, 0), dy = V(0, mx.y - mn.y, 0), dz = V(0, 0, mx.z - mn.z);
    float rad = radians(degrees), c = cosf(rad), sn = sinf(rad);
    v3 Qs[6] = {V(mn.x, mn.y, mx.z), V(mx.x, mn.y, mx.z), V(mx.x, mn.y, mn.z), V(mn.x, mn.y, mn.z), V(mn.x, mx.y, mx.z), V(mn.x, mn.y, mn.z)};
    v3 us[6] = {dx, neg(dz), neg(dx), dz, dx, dx};
    v3 vs[6] = {dy, dy, dy, dy, neg(dz), dz};
    for (int k = 0; k < 6; k++) cornell_quad(quads, n, add(rot_y(Qs[k], c, sn), offset), rot_y(us[k], c, sn), rot_y(vs[k], c, sn), mat);
}
orc_scene* orc_scene_cornell_box(void) {
    orc_material mats[4]; memset(mats, 0, sizeof(mats));
    st3(mats[0].albedo, V(0.65f, 0.05f, 0.05f)); mats[0].type = 0;  /* red */
    st3(mats[1].albedo, V(0.73f, 0.73f, 0.73f)); mats[1].type = 0;  /* white */
    st3(mats[2].albedo, V(0.12f, 0.45f, 0.15f)); mats[2].type = 0;  /* green */
    st3(mats[3].albedo, V(15.0f, 15.0f, 15.0f)); mats[3].type = 4;  /* light */
    orc_quad quads[18]; size_t n = 0;
    cornell_quad(quads, &n, V(555, 0, 0), V(0, 555, 0), V(0, 0, 555), 2);
    cornell_quad(quads, &n, V(0, 0, 0), V(0, 555, 0), V(0, 0, 555), 0);
    cornell_quad(quads, &n, V(343, 554, 332), V(-130, 0, 0), V(0, 0, -105), 3);
    cornell_quad(quads, &n, V(0, 0, 0), V(555, 0, 0), V(0, 0, 555), 1);
    cornell_quad(quads, &n, V(555, 555, 555), V(-555, 0, 0), V(0, 0, -555), 1);
    cornell_quad(quads, &n, V(0, 0, 555), V(555, 0, 0), V(0, 555, 0), 1);
    cornell_box(quads, &n, V(0, 0, 0), V(165, 330, 165), 15.0f, V(265, 0, 295), 1);
    cornell_box(quads, &n, V(0, 0, 0), V(165, 165, 165), -18.0f, V(130, 0, 65), 1);
    float black[3] = {0, 0, 0};
    return orc_scene_from_arrays_ext(0, NULL, n, quads, 4, mats, 0, 1, black);
}
/* final_scene() of "Ray Tracing: The Next Week" (BASELINE.json configs[4], nothing of it in the reference); the synthetic
 * planet replaces earthmap.jpg (integer arithmetic only).  Draw order: one uniform per ground box, then three per small
 * sphere, from the host stream. */
void orc_scene_set_perlin(orc_scene* s, uint64_t seed);
void orc_scene_set_image(orc_scene* s, uint32_t width, uint32_t height, const uint8_t* rgb);
static void synthetic_earth(uint8_t* img, uint32_t w, uint32_t h) {
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            uint32_t f = (x * x / 64u + y * y / 32u + x * y / 128u + 3u * x) % 64u;
            int land = f < 26u, ice = y < 9u || y > 118u;
            uint8_t* px = &img[((size_t)y * w + x) * 3];
            if (ice) { px[0] = 235; px[1] = 240; px[2] = 245; }
            else if (land) { px[0] = (uint8_t)(60u + 2u * f); px[1] = (uint8_t)(120u + f); px[2] = 50; }
            else { px[0] = 25; px[1] = (uint8_t)(60u + f / 2u); px[2] = (uint8_t)(140u + f); }
        }
}
static void b2_sphere(orc_prim* prims, size_t* n, v3 c0, v3 c1, float r, uint32_t mat, int moving) {
    orc_prim* p = &prims[(*n)++];
    st3(p->c0, c0); st3(p->c1, c1); p->radius = r; p->mat = mat | (moving ? ORC_PRIM_MOVING : 0u);
}
static void b2_mat(orc_material* m, uint32_t type, v3 albedo, float param) {
    memset(m, 0, sizeof(*m));
    st3(m->albedo, albedo); m->param = param; m->type = type;
}
orc_scene* orc_scene_book2_final(uint64_t seed) {
    orc_prim* prims = (orc_prim*)calloc(1008, sizeof(orc_prim));
    orc_quad* quads = (orc_quad*)calloc(2401, sizeof(orc_quad));
    orc_material mats[10];
    size_t np = 0, nq = 0;
    rng_t g; rng_init(&g, seed, 0u, 0u, 0x5CE9E5u);
    b2_mat(&mats[0], 0, V(0.48f, 0.83f, 0.53f), 0.0f);          /* ground */
    for (int i = 0; i < 20; i++)
        for (int j = 0; j < 20; j++) {
            const float w = 100.0f;
            float x0 = -1000.0f + (float)i * w, z0 = -1000.0f + (float)j * w, y0 = 0.0f;
            float x1 = x0 + w, y1 = 1.0f + 100.0f * rng_next(&g), z1 = z0 + w;
            cornell_box(quads, &nq, V(x0, y0, z0), V(x1, y1, z1), 0.0f, V(0, 0, 0), 0);
        }
    b2_mat(&mats[1], 4, V(7.0f, 7.0f, 7.0f), 0.0f);              /* light */
    cornell_quad(quads, &nq, V(123, 554, 147), V(300, 0, 0), V(0, 0, 265), 1);
    b2_mat(&mats[2], 0, V(0.7f, 0.3f, 0.1f), 0.0f);
    b2_sphere(prims, &np, V(400, 400, 200), V(430, 400, 200), 50.0f, 2, 1);
    b2_mat(&mats[3], 2, V(1.0f, 1.0f, 1.0f), 1.5f);              /* glass */
    b2_sphere(prims, &np, V(260, 150, 45), V(260, 150, 45), 50.0f, 3, 0);
    b2_mat(&mats[4], 1, V(0.8f, 0.8f, 0.9f), 1.0f);
    b2_sphere(prims, &np, V(0, 150, 145), V(0, 150, 145), 50.0f, 4, 0);
    b2_sphere(prims, &np, V(360, 150, 145), V(360, 150, 145), 70.0f, 3, 0);
    b2_mat(&mats[5], 5, V(0.2f, 0.4f, 0.9f), 0.2f);              /* blue medium inside the glass ball */
    b2_sphere(prims, &np, V(360, 150, 145), V(360, 150, 145), 70.0f, 5, 0);
    b2_mat(&mats[6], 5, V(1.0f, 1.0f, 1.0f), 0.0001f);           /* global fog */
    b2_sphere(prims, &np, V(0, 0, 0), V(0, 0, 0), 5000.0f, 6, 0);
    b2_mat(&mats[7], 7, V(1.0f, 1.0f, 1.0f), 0.0f);              /* image texture */
    b2_sphere(prims, &np, V(400, 200, 400), V(400, 200, 400), 100.0f, 7, 0);
    b2_mat(&mats[8], 6, V(0.5f, 0.5f, 0.5f), 0.2f);              /* marble */
    b2_sphere(prims, &np, V(220, 280, 300), V(220, 280, 300), 80.0f, 8, 0);
    b2_mat(&mats[9], 0, V(0.73f, 0.73f, 0.73f), 0.0f);
    float rad = radians(15.0f), c = cosf(rad), sn = sinf(rad);
    for (int j = 0; j < 1000; j++) {
        v3 ctr;
        ctr.x = 165.0f * rng_next(&g); ctr.y = 165.0f * rng_next(&g); ctr.z = 165.0f * rng_next(&g);
        v3 pos = add(rot_y(ctr, c, sn), V(-100, 270, 395));
        b2_sphere(prims, &np, pos, pos, 10.0f, 9, 0);
    }
    float black[3] = {0, 0, 0};
    orc_scene* s = orc_scene_from_arrays_ext(np, prims, nq, quads, 10, mats, 0, 1, black);
    free(prims); free(quads);
    uint8_t* img = (uint8_t*)malloc(256 * 128 * 3);
    synthetic_earth(img, 256, 128);
    orc_scene_set_image(s, 256, 128, img);
    free(img);
    orc_scene_set_perlin(s, seed);
    return s;
}
void orc_scene_world(const orc_scene* s, orc_world* out) { *out = s->world; }
/* perlin::perlin() of "The Next Week": randvec[i] = unit_vector(random(-1,1)^3); perm = identity shuffled by
 * `for i = n-1 .. 1: swap(p[i], p[random_int(0, i)])`, three times; uniforms from the host stream, id 0x9E81 */
void orc_scene_set_perlin(orc_scene* s, uint64_t seed) {
    rng_t g; rng_init(&g, seed, 0u, 0u, 0x9E81u);
    if (!s->perlin) s->perlin = (orc_perlin*)malloc(sizeof(orc_perlin));
    orc_perlin* t = s->perlin;
    for (int i = 0; i < 256; i++) {
        v3 v;
        v.x = rng_next(&g) * 2.0f - 1.0f;
        v.y = rng_next(&g) * 2.0f - 1.0f;
        v.z = rng_next(&g) * 2.0f - 1.0f;
        if (near_zero(v)) v = V(1.0f, 0.0f, 0.0f);
        st3(t->randvec[i], normalize(v));
    }
    for (int k = 0; k < 3; k++) {
        for (int i = 0; i < 256; i++) t->perm[k][i] = i;
        for (int i = 255; i > 0; i--) {
            int target = (int)(rng_next(&g) * (float)(i + 1));
            if (target > i) target = i;
            int32_t tmp = t->perm[k][i]; t->perm[k][i] = t->perm[k][target]; t->perm[k][target] = tmp;
        }
    }
    s->world.perlin = t;
}
void orc_scene_set_image(orc_scene* s, uint32_t width, uint32_t height, const uint8_t* rgb) {
    free(s->image);
    s->image = (uint8_t*)malloc((size_t)width * height * 3);
    memcpy(s->image, rgb, (size_t)width * height * 3);
    s->image_w = width; s->image_h = height;
    s->world.image = s->image; s->world.image_width = width; s->world.image_height = height;
}
void orc_scene_free(orc_scene* s) {
    if (!s) return;
    free(s->prims); free(s->quads); free(s->mats); free(s->nodes); free(s->perlin); free(s->image); free(s);
}

/* ---- twins of the fixtures oracle/ref_path_probe.cpp generates from the reference's own headers (tests/golden/ref_*) ---- */
/* aabb::longest_axis / surface_area / centeroid / union ctor / operator+= / box_{x,y,z}_compare (aabb.cuh:19,24,46-68,78-88):
 * boxes n*12 (a.min a.max b.min b.max) -> out n*20 [axis, area, centroid 3, union min 3 max 3, a+=b min 3 max 3, cmp x y z] */
void orc_aabb_misc_batch(size_t n, const float* boxes, float* out) {
    for (size_t i = 0; i < n; i++) {
        box_t a = {ld3(boxes + 12 * i), ld3(boxes + 12 * i + 3)}, b = {ld3(boxes + 12 * i + 6), ld3(boxes + 12 * i + 9)};
        float* o = out + 20 * i;
        o[0] = (float)box_longest_axis(a);
        o[1] = box_surface_area(a);
        st3(o + 2, box_centroid(a));
        box_t u = box_union(a, b);
        st3(o + 5, u.mn); st3(o + 8, u.mx);
        box_t p = a; p = box_union(p, b);   /* operator+= is the same min/max with the operands in the same order */
        st3(o + 11, p.mn); st3(o + 14, p.mx);
        o[17] = a.mn.x < b.mn.x ? 1.0f : 0.0f;
        o[18] = a.mn.y < b.mn.y ? 1.0f : 0.0f;
        o[19] = a.mn.z < b.mn.z ? 1.0f : 0.0f;
    }
}
/* checker_texture(c1, c2, scale)::value (cu_Textures.cuh:26-39): in n*10 (even 3, odd 3, scale, pos 3) -> colour n*3 */
void orc_checker_batch(size_t n, const float* in, float* out) {
    for (size_t i = 0; i < n; i++) {
        orc_material m;
        memset(&m, 0, sizeof(m));
        memcpy(m.albedo, in + 10 * i, 12); memcpy(m.albedo2, in + 10 * i + 3, 12);
        m.param = 1.0f / in[10 * i + 6];   /* inv_scale(1.0f / scale), cu_Textures.cuh:27 */
        st3(out + 3 * i, checker_value(&m, ld3(in + 10 * i + 7)));
    }
}
/* Ray::at (ray_data.cuh:14) and isBackfacing (ray_data.cuh:44-46): in n*10 (o, d, t, normal) -> out n*4 (at 3, backfacing) */
void orc_ray_batch(size_t n, const float* in, float* out) {
    for (size_t i = 0; i < n; i++) {
        ray_t r; r.o = ld3(in + 10 * i); r.d = ld3(in + 10 * i + 3); r.time = 0.0f;
        st3(out + 4 * i, ray_at(&r, in[10 * i + 6]));
        out[4 * i + 3] = dot(r.d, ld3(in + 10 * i + 7)) > 0 ? 1.0f : 0.0f;
    }
}
/* leaf tests / box tests of one trace per ray (the instrumented counters of the render loop, per ray) */
int orc_trace_counts(const orc_world* w, size_t n, const float* rays, uint32_t* out_leaf_tests, uint32_t* out_box_tests) {
    int err = 0;
    for (size_t i = 0; i < n; i++) {
        ray_t r = ld_ray7(rays + 7 * i);
        rec_t rec;
        memset(&rec, 0, sizeof(rec));
        rec.distance = ORC_MISS_DIST; rec.prim = -1;
        orc_counters c;
        memset(&c, 0, sizeof(c));
        rng_t g;
        rng_init(&g, 0u, (uint32_t)i, 0u, 0x7ACEu);
        world_closest_intersection(w, &r, &rec, &c, &err, &g);
        out_leaf_tests[i] = (uint32_t)c.leaf_tests; out_box_tests[i] = (uint32_t)c.box_tests;
    }
    return err;
}

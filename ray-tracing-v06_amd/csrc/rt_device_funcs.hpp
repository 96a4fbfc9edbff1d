// rt_device_funcs.hpp — the hot-path functions as flat, non-virtual device code.
//
// The reference dispatches Hittable::ClosestIntersection and Material::Scatter through device
// vtables (rt_engine/geometry/hittable.cuh:18, rt_engine/shaders/material.cuh:24-28) and chases
// hittables[i] -> SphereHittable{vptr,sphere*,mat*} -> Sphere (3 dependent loads per leaf).  Here
// the world is three linear arrays and every call is a switch on a tag.  Arithmetic follows the
// reference expression by expression (rt_math.hpp) so the CPU oracle and the GPU agree bit for bit.
#pragma once
#include "rt06.h"
#include "rt_internal.hpp"
#include "rt_math.hpp"

struct DeviceWorld {
    uint32_t kind;
    int32_t root;
    uint32_t n_nodes, n_prims, n_mats;
    f3 bmin, bmax;
    const rt_bvh_node* nodes;
    const rt_prim* prims;
    const rt_material* mats;
    // extension beyond the reference: quads (primitive index >= n_prims), constant background
    const rt_quad* quads;
    uint32_t n_quads;
    uint32_t background;
    f3 background_color;
    const rt_perlin* perlin;   // RT_MAT_LAMBERTIAN_NOISE tables (global memory), or null
    const uint8_t* image;      // RT_MAT_LAMBERTIAN_IMAGE RGB8 image, or null
    uint32_t image_w, image_h;
    uint32_t traversal;        // RT_TRAVERSAL_STACK / RT_TRAVERSAL_QUEUE (BVH worlds)
    uint32_t* error_flag;      // set to 1 when the distance-sorted queue overflows its 32 entries (checked by the host, never silent)
};

// RayPayload (ray_data.cuh:33-40) with Sphere::TraceRecord (SphereHittable.cuh:38-41) unpacked
struct HitRec {
    float distance;
    f3 normal;
    int32_t prim;
    uint32_t mat;
};

// aabb::intersects, rt_engine/geometry/aabb.cuh:30-44
RT_HD bool aabb_intersects(f3 box_min, f3 box_max, const Ray& ray, float ray_max_dist, float& dist) {
    f3 bmin = (box_min - ray.o) / ray.d;
    f3 bmax = (box_max - ray.o) / ray.d;
    f3 tmp_min = glm_min(bmin, bmax);
    bmax = glm_max(bmin, bmax);
    bmin = tmp_min;
    float tmin = comp_max(bmin);
    float tmax = comp_min(bmax);
    bool hit = tmin <= tmax && tmin < ray_max_dist && tmax > 0;
    if (hit) dist = tmin;
    return hit;
}

// _sphere_closest_intersection, rt_engine/geometry/SphereHittable.cuh:15-33
// `a` = dot(ray.d, ray.d) is a property of the ray: the streaming kernel computes it once per trace (same expression, same bits)
RT_HD float sphere_closest_intersection_a(const Ray& ray, float a, f3 center, float radius);
RT_HD float sphere_closest_intersection(const Ray& ray, f3 center, float radius) {
    return sphere_closest_intersection_a(ray, dot(ray.d, ray.d), center, radius);
}
RT_HD float sphere_closest_intersection_a(const Ray& ray, float a, f3 center, float radius) {
    f3 oc = ray.o - center;
    float hb = dot(ray.d, oc);
    float c = dot(oc, oc) - radius * radius;
    float d = hb * hb - a * c;
    if (d <= 0) return RT_MISS_DIST;
    d = sqrtf(d);
    float t = (-hb - d) / a;
    if (t < 0.0f) {
        t = (-hb + d) / a;
        if (t < 0.0f) return RT_MISS_DIST;
    }
    return t;
}

// SphereHittable::ClosestIntersection (SphereHittable.cu:56-66) /
// MovingSphereHittable::ClosestIntersection (:91-102)
// constant_medium::hit of "The Next Week" (extension, not in the reference): a sphere whose material is RT_MAT_ISOTROPIC
// IS a constant medium bounded by that sphere, density = the material's param.  Reference conventions: the ray interval
// is [0, rec.distance) instead of [t_min, t_max], a tangent ray misses (SphereHittable.cuh:22).  ONE uniform is drawn
// per test that finds a non-empty interval inside the boundary, in traversal order.  Returns the hit distance or
// RT_MISS_DIST.
RT_HD float medium_sphere_intersection(const Ray& ray, f3 center, float radius, float density, float max_dist, Rng& rng) {
    f3 oc = ray.o - center;
    float a = dot(ray.d, ray.d);
    float hb = dot(ray.d, oc);
    float c = dot(oc, oc) - radius * radius;
    float d = hb * hb - a * c;
    if (d <= 0) return RT_MISS_DIST;
    d = sqrtf(d);
    float t1 = (-hb - d) / a, t2 = (-hb + d) / a;
    float t_in = t1 < 0.0f ? 0.0f : t1;
    float t_out = t2 > max_dist ? max_dist : t2;
    if (!(t_in < t_out)) return RT_MISS_DIST;
    float ray_length = sqrtf(a);
    float distance_inside = (t_out - t_in) * ray_length;
    float hit_distance = (-1.0f / density) * rt_logf(rng.next());
    if (hit_distance > distance_inside) return RT_MISS_DIST;
    return t_in + hit_distance / ray_length;
}

RT_HD bool prim_closest_intersection(const rt_prim& p, int32_t idx, const Ray& ray, HitRec& rec, const rt_material* mats, Rng* rng) {
    f3 center = mk3(p.c0[0], p.c0[1], p.c0[2]);
    if (p.mat & RT_PRIM_MOVING) center = mix(center, mk3(p.c1[0], p.c1[1], p.c1[2]), ray.time);
    const rt_material& pm = mats[p.mat & ~RT_PRIM_MOVING];
    if (pm.type == RT_MAT_ISOTROPIC) {
        float tm = medium_sphere_intersection(ray, center, p.radius, pm.param, rec.distance, *rng);
        if (tm >= rec.distance) return false;
        rec.mat = p.mat & ~RT_PRIM_MOVING;
        rec.distance = tm;
        rec.prim = idx;
        rec.normal = mk3(1.0f, 0.0f, 0.0f);  // arbitrary, as in the book: an isotropic scatter ignores it
        return true;
    }
    float t = sphere_closest_intersection(ray, center, p.radius);
    if (t >= rec.distance) return false;
    rec.mat = p.mat & ~RT_PRIM_MOVING;
    rec.distance = t;
    rec.prim = idx;
    rec.normal = (ray_at(ray, rec.distance) - center) / p.radius;
    return true;
}

// quad::hit of "Ray Tracing: The Next Week" in the reference's conventions: t >= 0 accepted (no t_min: the origin is
// offset instead, Renderer.cu:175), `t >= rec.distance` rejects; the stored normal faces AGAINST the ray (two-sided).
RT_HD bool quad_closest_intersection(f3 Q, float D, f3 u, f3 v, f3 n, f3 w, uint32_t mat, int32_t unified, const Ray& ray, HitRec& rec) {
    float denom = dot(n, ray.d);
    if (fabsf(denom) < 1e-8f) return false;
    float t = (D - dot(n, ray.o)) / denom;
    if (t < 0.0f) return false;
    if (t >= rec.distance) return false;
    f3 planar = ray_at(ray, t) - Q;
    float alpha = dot(w, cross(planar, v));
    float beta = dot(w, cross(u, planar));
    if (!(alpha >= 0.0f && alpha <= 1.0f && beta >= 0.0f && beta <= 1.0f)) return false;
    rec.mat = mat;
    rec.distance = t;
    rec.prim = unified;
    rec.normal = (dot(ray.d, n) > 0) ? -n : n;
    return true;
}

__device__ inline bool any_prim_closest_intersection(const DeviceWorld& w, int32_t idx, const Ray& ray, HitRec& rec, Rng* rng) {
    if ((uint32_t)idx >= w.n_prims) {
        const rt_quad& q = w.quads[(uint32_t)idx - w.n_prims];
        return quad_closest_intersection(mk3(q.Q[0], q.Q[1], q.Q[2]), q.D, mk3(q.u[0], q.u[1], q.u[2]), mk3(q.v[0], q.v[1], q.v[2]),
                                         mk3(q.normal[0], q.normal[1], q.normal[2]), mk3(q.w[0], q.w[1], q.w[2]), q.mat, idx, ray, rec);
    }
    return prim_closest_intersection(w.prims[idx], idx, ray, rec, w.mats, rng);
}

RT_HD bool node_box(const rt_bvh_node& n, const Ray& ray, float maxd, float& dist) {
    return aabb_intersects(mk3(n.min[0], n.min[1], n.min[2]), mk3(n.max[0], n.max[1], n.max[2]), ray, maxd, dist);
}

// BVH::ClosestIntersection, rt_engine/geometry/BVH.cu:54-106 (the live, non-priority-queue branch):
// root box first; pop; leaf -> primitive; inner -> test BOTH child boxes, order near-first, push far
// then near iff dist < rec.distance; no re-check at pop.
__device__ inline bool bvh_closest_intersection(const DeviceWorld& w, const Ray& ray, HitRec& rec, Rng* rng) {
    int32_t stack[RT_MAX_STACK];
    int head = 0;
    float root_dist;
    if (!node_box(w.nodes[w.root], ray, rec.distance, root_dist)) return false;
    stack[head++] = w.root;
    bool hit_any = false;
    while (head != 0) {
        int32_t idx = stack[--head];
        const rt_bvh_node& node = w.nodes[idx];
        int32_t left_idx = node.left, right_idx = node.right;
        if (left_idx == -1) {
            hit_any |= any_prim_closest_intersection(w, right_idx, ray, rec, rng);
            continue;
        }
        float left_dist = RT_MISS_DIST, right_dist = RT_MISS_DIST;
        node_box(w.nodes[left_idx], ray, rec.distance, left_dist);
        node_box(w.nodes[right_idx], ray, rec.distance, right_dist);
        if (left_dist > right_dist) {
            int32_t ti = left_idx; left_idx = right_idx; right_idx = ti;
            float tf = left_dist; left_dist = right_dist; right_dist = tf;
        }
        if (right_dist < rec.distance) stack[head++] = right_idx;
        if (left_dist < rec.distance) stack[head++] = left_idx;
    }
    return hit_any;
}

// BVH::ClosestIntersection with the reference's disabled distance-sorted queue (BVH.cu:17-49 and the `#if _USE_PRIO_QUEUE`
// branch :80-86): each hit child box is enqueued with its entry distance, the queue stays sorted with the nearest entry on
// top, the nearest entry of the whole frontier is dequeued next.  The reference's enqueue writes `distances[head]` AFTER
// `head++` (:37-39: index and distance end up in different slots); fixed here and in the oracle — same slot, then the
// insertion sort as written.  Capacity _PRIO_QUEUE_ELEM_COUNT = 32, checked (the reference does not): overflow raises
// *w.error_flag and ends the walk.  Probes, baseline kernel and the streaming kernel's RT_WORLD_BVH_QUEUE mode (see RT_TRAVERSAL_QUEUE in rt06.h).
__device__ inline bool bvh_closest_intersection_queue(const DeviceWorld& w, const Ray& ray, HitRec& rec, Rng* rng) {
    int32_t indices[RT_MAX_STACK];
    float distances[RT_MAX_STACK];
    int head = 0;
    float root_dist;
    if (!node_box(w.nodes[w.root], ray, rec.distance, root_dist)) return false;
    bool hit_any = false;
    auto enqueue = [&](int32_t idx, float dist) -> bool {
        if (head >= RT_MAX_STACK) { *w.error_flag = 1u; return false; }
        indices[head] = idx; distances[head] = dist; head++;
        for (int i = head - 1; i >= 1; i--) {
            if (distances[i] > distances[i - 1]) {
                float td = distances[i]; distances[i] = distances[i - 1]; distances[i - 1] = td;
                int32_t ti = indices[i]; indices[i] = indices[i - 1]; indices[i - 1] = ti;
            } else break;
        }
        return true;
    };
    enqueue(w.root, root_dist);
    while (head != 0) {
        const int32_t idx = indices[--head];
        const rt_bvh_node& node = w.nodes[idx];
        if (node.left == -1) {
            hit_any |= any_prim_closest_intersection(w, node.right, ray, rec, rng);
            continue;
        }
        float left_dist = RT_MISS_DIST, right_dist = RT_MISS_DIST;
        if (node_box(w.nodes[node.left], ray, rec.distance, left_dist) && !enqueue(node.left, left_dist)) return hit_any;
        if (node_box(w.nodes[node.right], ray, rec.distance, right_dist) && !enqueue(node.right, right_dist)) return hit_any;
    }
    return hit_any;
}

// A 4-wide walk of the same binary tree (RT_TRAVERSAL_WIDE4; SURVEY §8f rank 4 — the reference has binary nodes only, BVH.cuh:16-25): a visit
// looks two levels down — the children of the node's children, a leaf child standing for itself: up to four boxes —, tests each against
// rec.distance (BVH.cu:87-88), orders them nearest first (stable, a missed box keeps _MISS_DIST) and pushes them far-to-near iff
// dist < rec.distance (BVH.cu:95-96).  The oracle's bvh_closest_intersection_wide4, statement by statement.  Overflow of the 32 entries
// raises *w.error_flag and ends the walk (at most 3 * ceil(depth / 2) + 1 entries are ever needed).
__device__ inline bool bvh_closest_intersection_wide4(const DeviceWorld& w, const Ray& ray, HitRec& rec, Rng* rng) {
    int32_t stack[RT_MAX_STACK];
    int head = 0;
    float root_dist;
    if (!node_box(w.nodes[w.root], ray, rec.distance, root_dist)) return false;
    stack[head++] = w.root;
    bool hit_any = false;
    while (head != 0) {
        const int32_t idx = stack[--head];
        const rt_bvh_node& node = w.nodes[idx];
        if (node.left == -1) {
            hit_any |= any_prim_closest_intersection(w, node.right, ray, rec, rng);
            continue;
        }
        int32_t cand[4];
        float dist[4];
        int n = 0;
        const int32_t kids[2] = {node.left, node.right};
        for (int k = 0; k < 2; k++) {
            const rt_bvh_node& c = w.nodes[kids[k]];
            if (c.left == -1) cand[n++] = kids[k];
            else { cand[n++] = c.left; cand[n++] = c.right; }
        }
        for (int i = 0; i < n; i++) {
            dist[i] = RT_MISS_DIST;
            node_box(w.nodes[cand[i]], ray, rec.distance, dist[i]);
        }
        for (int i = 1; i < n; i++)
            for (int j = i; j >= 1 && dist[j - 1] > dist[j]; j--) {
                const float td = dist[j]; dist[j] = dist[j - 1]; dist[j - 1] = td;
                const int32_t ti = cand[j]; cand[j] = cand[j - 1]; cand[j - 1] = ti;
            }
        if (head + n > RT_MAX_STACK) { *w.error_flag = 1u; return hit_any; }
        for (int i = n - 1; i >= 0; i--)
            if (dist[i] < rec.distance) stack[head++] = cand[i];
    }
    return hit_any;
}

// HittableList::ClosestIntersection, rt_engine/geometry/HittableList.cuh:21-34
__device__ inline bool list_closest_intersection(const DeviceWorld& w, const Ray& ray, HitRec& rec, Rng* rng) {
    float d;
    if (!aabb_intersects(w.bmin, w.bmax, ray, rec.distance, d)) return false;
    bool hit_any = false;
    for (uint32_t i = 0; i < w.n_prims + w.n_quads; i++)
        if (any_prim_closest_intersection(w, (int32_t)i, ray, rec, rng)) hit_any = true;
    return hit_any;
}

// bvh_node::ClosestIntersection, rt_engine/geometry/bvh_node.cuh:19-24, made iterative: the recursion
// "own box, then left subtree, then right subtree" is a pre-order walk, i.e. pop / test / push right /
// push left with the box test at visit time against the current rec.distance.
__device__ inline bool tree_closest_intersection(const DeviceWorld& w, const Ray& ray, HitRec& rec, Rng* rng) {
    int32_t stack[RT_MAX_STACK];
    int head = 0;
    stack[head++] = w.root;
    bool hit_any = false;
    while (head != 0) {
        int32_t ref = stack[--head];
        if (ref < 0) {
            int32_t pi = -ref - 1;
            hit_any |= prim_closest_intersection(w.prims[pi], pi, ray, rec, w.mats, rng);
            continue;
        }
        const rt_bvh_node& n = w.nodes[ref];
        float d;
        if (!node_box(n, ray, rec.distance, d)) continue;
        stack[head++] = n.right;
        stack[head++] = n.left;
    }
    return hit_any;
}

// rng: drawn from by constant media only (one uniform per test that enters the boundary)
__device__ inline bool world_closest_intersection(const DeviceWorld& w, const Ray& ray, HitRec& rec, Rng* rng) {
    if (w.kind == RT_WORLD_BVH)
        return w.traversal == RT_TRAVERSAL_QUEUE ? bvh_closest_intersection_queue(w, ray, rec, rng)
             : w.traversal == RT_TRAVERSAL_WIDE4 ? bvh_closest_intersection_wide4(w, ray, rec, rng) : bvh_closest_intersection(w, ray, rec, rng);
    if (w.kind == RT_WORLD_LIST) return list_closest_intersection(w, ray, rec, rng);
    return tree_closest_intersection(w, ray, rec, rng);
}

// reflectance, rt_engine/shaders/cu_materials.cuh:99-104.  powf(1-cos,5) is evaluated as
// x2=x*x; x4=x2*x2; x5=x4*x on BOTH sides of the parity check: libm, OCML and CUDA powf differ in
// the last ulp, and that ulp decides `reflect_prob > rng.next()`.
RT_HD float reflectance(float cos_theta, float ior_ratio) {
    float r0 = (1 - ior_ratio) / (1 + ior_ratio);
    r0 = r0 * r0;
    float x = 1 - cos_theta;
    float x2 = x * x;
    float x4 = x2 * x2;
    float x5 = x4 * x;
    return r0 + (1 - r0) * x5;
}

// checker_texture::value, rt_engine/shaders/cu_Textures.cuh:31-39 (ivec3 truncates toward zero)
RT_HD f3 checker_value(f3 even, f3 odd, float inv_scale, f3 pos) {
    f3 sp = pos * inv_scale;
    int ix = (int)sp.x, iy = (int)sp.y, iz = (int)sp.z;
    int sum = 0;
    sum += ix; sum += iy; sum += iz;
    bool is_even = (sum % 2 == 0);
    return mk3(is_even ? even.x : odd.x, is_even ? even.y : odd.y, is_even ? even.z : odd.z);
}

// perlin::noise / perlin_interp / turb and noise_texture::value of "The Next Week" (extension, not in the reference), fp32
// in one fixed evaluation order; the sine is rt_sinf.
RT_HD float perlin_noise(const rt_perlin* t, f3 p) {
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
RT_HD float perlin_turb(const rt_perlin* t, f3 p, int depth) {
    float accum = 0.0f, weight = 1.0f;
    for (int i = 0; i < depth; i++) {
        accum += weight * perlin_noise(t, p);
        weight *= 0.5f;
        p = p * 2.0f;
    }
    return fabsf(accum);
}
RT_HD f3 noise_value(const rt_perlin* t, f3 albedo, float scale, f3 p) {
    return albedo * (1.0f + rt_sinf(scale * p.z + 10.0f * perlin_turb(t, p, 7)));
}
// sphere::get_sphere_uv + image_texture::value of "The Next Week" (extension): n = outward unit normal of the sphere
RT_HD f3 image_texel(const uint8_t* image, uint32_t width, uint32_t height, float u, float v);
RT_HD f3 image_value(const uint8_t* image, uint32_t width, uint32_t height, f3 n) {
    float cy = -n.y;
    cy = cy < -1.0f ? -1.0f : (cy > 1.0f ? 1.0f : cy);   // the normal is unit only up to rounding
    float theta = rt_acosf(cy);
    float phi = rt_atan2f(-n.z, n.x) + 0x1.921fb6p+1f;
    return image_texel(image, width, height, phi / 0x1.921fb6p+2f, theta / 0x1.921fb6p+1f);
}
// on a quad the texture coordinates are the planar coordinates (alpha, beta) of quad::hit, recomputed from the hit point
RT_HD f3 image_value_quad(const uint8_t* image, uint32_t width, uint32_t height, f3 Q, f3 u, f3 v, f3 w, f3 hit_p) {
    f3 planar = hit_p - Q;
    float alpha = dot(w, cross(planar, v));
    float beta = dot(w, cross(u, planar));
    return image_texel(image, width, height, alpha, beta);
}
// image_texture::value(u, v) of "The Next Week": clamp, flip v to image coordinates, nearest texel, 1/255
RT_HD f3 image_texel(const uint8_t* image, uint32_t width, uint32_t height, float u, float v) {
    u = u < 0.0f ? 0.0f : (u > 1.0f ? 1.0f : u);
    v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
    v = 1.0f - v;   // flip to image coordinates
    int i = (int)(u * (float)width), j = (int)(v * (float)height);
    if (i > (int)width - 1) i = (int)width - 1;
    if (j > (int)height - 1) j = (int)height - 1;
    const uint8_t* px = image + ((size_t)j * width + (size_t)i) * 3u;
    return mk3((float)px[0] * 0x1.010102p-8f, (float)px[1] * 0x1.010102p-8f, (float)px[2] * 0x1.010102p-8f);
}

// Material::Scatter — LambertianAbstract (cu_materials.cuh:52-64), MetalAbstract (:77-95),
// DielectricAbstract (:115-143), LambertianTexture (:27-40)
// `w`: the world's Perlin tables / image for the two textured extension materials (may be null otherwise)
RT_HD bool material_scatter(const rt_material& m, const Ray& in_ray, const HitRec& rec, Rng& rng, Ray& out, f3& attenuation,
                            const DeviceWorld* w = nullptr) {
    f3 normal = rec.normal;
    if (m.type == RT_MAT_DIFFUSE_LIGHT) return false;  // diffuse_light of "The Next Week": emits, never scatters
    const f3 albedo = mk3(m.albedo[0], m.albedo[1], m.albedo[2]);
    const f3 albedo2 = mk3(m.albedo2[0], m.albedo2[1], m.albedo2[2]);
    if (m.type == RT_MAT_LAMBERTIAN || m.type == RT_MAT_LAMBERTIAN_CHECKER || m.type == RT_MAT_LAMBERTIAN_NOISE || m.type == RT_MAT_LAMBERTIAN_IMAGE) {
        f3 ray_dir = normal + rng_on_unit3(rng);
        if (near_zero(ray_dir)) return false;
        out.o = ray_at(in_ray, rec.distance); out.d = ray_dir; out.time = in_ray.time;
        if (m.type == RT_MAT_LAMBERTIAN) attenuation = albedo;
        else if (m.type == RT_MAT_LAMBERTIAN_CHECKER) attenuation = checker_value(albedo, albedo2, m.param, ray_at(in_ray, rec.distance));
        else if (m.type == RT_MAT_LAMBERTIAN_NOISE) attenuation = noise_value(w->perlin, albedo, m.param, ray_at(in_ray, rec.distance));
        else if (rec.prim >= 0 && (uint32_t)rec.prim >= w->n_prims) {
            const rt_quad& q = w->quads[(uint32_t)rec.prim - w->n_prims];
            attenuation = image_value_quad(w->image, w->image_w, w->image_h, mk3(q.Q[0], q.Q[1], q.Q[2]), mk3(q.u[0], q.u[1], q.u[2]), mk3(q.v[0], q.v[1], q.v[2]),
                                           mk3(q.w[0], q.w[1], q.w[2]), ray_at(in_ray, rec.distance));
        } else attenuation = image_value(w->image, w->image_w, w->image_h, normal);
        return true;
    }
    if (m.type == RT_MAT_ISOTROPIC) {  // isotropic phase function of "The Next Week": a uniformly random direction, always scatters
        out.o = ray_at(in_ray, rec.distance); out.d = rng_on_unit3(rng); out.time = in_ray.time;
        attenuation = albedo;
        return true;
    }
    if (m.type == RT_MAT_METAL) {
        f3 scatter_dir = reflect(in_ray.d, normal) + rng_on_unit3(rng) * m.param;
        if (dot(scatter_dir, normal) < 0 || near_zero(scatter_dir)) return false;
        out.o = ray_at(in_ray, rec.distance); out.d = scatter_dir; out.time = in_ray.time;
        attenuation = albedo;
        return true;
    }
    float ior = m.param;
    bool hit_backface = dot(in_ray.d, normal) > 0;  // isBackfacing, ray_data.cuh:44-46
    if (hit_backface) normal = -normal;
    float ior_ratio = hit_backface ? ior : 1 / ior;
    f3 unit_dir = normalize(in_ray.d);
    float cos_theta = fminf(dot(-unit_dir, normal), 1.0f);
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    float reflect_prob = reflectance(cos_theta, ior_ratio);
    f3 scatter_dir;
    if (ior_ratio * sin_theta > 1.0f || reflect_prob > rng.next()) scatter_dir = reflect(unit_dir, normal);
    else scatter_dir = refract(unit_dir, normal, ior_ratio);
    out.o = ray_at(in_ray, rec.distance); out.d = scatter_dir; out.time = in_ray.time;
    attenuation = albedo;
    return true;
}

// sample_ray — PinholeCamera (cu_Cameras.cuh:27-30), DefocusBlurCamera (:54-64), MotionBlurCamera (:87-89), in two halves:
// camera_draw consumes the camera's uniforms (defocus: the lens point InUnit<2>; motion: one uniform -> time), camera_ray
// is the arithmetic.  The streaming path runs the first half in primary_rays_kernel and the second at regeneration.
RT_HD void camera_draw(const rt_camera& c, Rng& rng, float& a, float& b) {
    a = 0.0f; b = 0.0f;
    if (c.type == RT_CAM_DEFOCUS) rng_in_unit2(rng, a, b);                // a, b = lens point in the unit disc
    else if (c.type == RT_CAM_MOTION) a = mix(c.t0, c.t1, rng.next());  // a = ray time
}
RT_HD Ray camera_ray(const rt_camera& c, float s, float t, float a, float b) {
    Ray r;
    f3 o = mk3(c.o[0], c.o[1], c.o[2]), u = mk3(c.u[0], c.u[1], c.u[2]);
    f3 v = mk3(c.v[0], c.v[1], c.v[2]), w = mk3(c.w[0], c.w[1], c.w[2]);
    if (c.type == RT_CAM_DEFOCUS) {
        f3 offset = u * a + v * b;
        offset = offset * c.lens_radius;
        f3 forward = w * c.focus_dist;
        f3 hori = u * c.viewport_width * c.focus_dist;
        f3 vert = v * c.viewport_height * c.focus_dist;
        r.o = o + offset;
        r.d = forward + hori * s + vert * t - offset;
        r.time = 0.0f;
    } else {
        r.o = o;
        r.d = w + u * s + v * t;
        r.time = a;   // 0 for a pinhole camera
    }
    return r;
}
RT_HD Ray camera_sample_ray(const rt_camera& c, float s, float t, Rng& rng) {
    float a, b;
    camera_draw(c, rng, a, b);
    return camera_ray(c, s, t, a, b);
}

// sample_world, main/src/Renderer.cu:139-181.  The emission / background hooks are the reference's own commented
// placeholders (`accum_radiance`, Renderer.cu:142,152,157,163,179); with no emissive material and background 0 this
// is exactly the live function.
__device__ inline f3 sample_world(const DeviceWorld& w, Ray cur_ray, uint32_t max_depth, Rng& rng) {
    f3 accum_attenuation = mk3(1.0f);
    f3 accum_radiance = mk3(0.0f);
    for (uint32_t i = 0; i < max_depth; i++) {
        HitRec rec;
        rec.distance = RT_MISS_DIST; rec.normal = mk3(0.0f); rec.prim = -1; rec.mat = 0;
        if (!world_closest_intersection(w, cur_ray, rec, &rng)) {
            f3 sky;
            if (w.background == 1u) {
                sky = w.background_color;
            } else {
                float t = normalize(cur_ray.d).y * 0.5f + 0.5f;
                sky = linear_interpolate(mk3(0.1f, 0.2f, 0.4f), mk3(0.9f, 0.9f, 0.99f), t);
            }
            return accum_attenuation * sky + accum_radiance;
        }
        const rt_material& m = w.mats[rec.mat];
        if (m.type == RT_MAT_DIFFUSE_LIGHT) accum_radiance = accum_radiance + accum_attenuation * mk3(m.albedo[0], m.albedo[1], m.albedo[2]);
        Ray scattered;
        f3 attenuation;
        if (!material_scatter(m, cur_ray, rec, rng, scattered, attenuation, &w)) return accum_radiance;
        accum_attenuation = accum_attenuation * attenuation;
        cur_ray = scattered;
        cur_ray.o = cur_ray.o + cur_ray.d * 0.001f;
    }
    return accum_radiance;
}

// pixel centre in NDC, Renderer.cu:188-192
RT_HD void pixel_ndc(uint32_t x, uint32_t y, uint32_t width, uint32_t height, float& psx, float& psy, float& ndcx, float& ndcy) {
    psx = 1.0f / (float)width;
    psy = 1.0f / (float)height;
    ndcx = ((float)x + 0.5f) * psx * 2.0f - 1.0f;
    ndcy = ((float)y + 0.5f) * psy * 2.0f - 1.0f;
}

// camera ray of one sample: jitter in a disc of half a pixel, then sample_ray (Renderer.cu:199-201).
// The RNG is keyed per (pixel, sample) — SURVEY.md Appendix A item 7.
RT_HD Ray primary_ray(const rt_camera& cam, uint32_t width, uint32_t height, uint32_t gid, Rng& rng) {
    uint32_t x = gid % width, y = gid / width;
    float psx, psy, ndcx, ndcy;
    pixel_ndc(x, y, width, height, psx, psy, ndcx, ndcy);
    float jx, jy;
    rng_in_unit2(rng, jx, jy);
    return camera_sample_ray(cam, ndcx + jx * psx, ndcy + jy * psy, rng);
}

__device__ inline f3 one_sample(const DeviceWorld& w, const rt_camera& cam, uint32_t width, uint32_t height,
                                uint32_t max_depth, uint64_t seed, uint32_t gid, uint32_t s) {
    Rng rng;
    rng.init(seed, gid, s, RT_STREAM_RENDER);
    Ray ray = primary_ray(cam, width, height, gid, rng);
    return sample_world(w, ray, max_depth, rng);
}

// ref_path_probe.cpp — golden-vector generator for the part of the hot path ABOVE the GLM vocabulary (test infrastructure).
//
// Compiles the REFERENCE's own headers from where they lie under /root/reference (never copied):
//     main/src/rt_engine/ray_data.cuh                 Ray, RayPayload, isBackfacing
//     main/src/rt_engine/geometry/aabb.cuh            aabb::intersects, longest_axis, surface_area, centeroid, union, box_*_compare
//     main/src/rt_engine/geometry/hittable.cuh        Hittable (abstract)
//     main/src/rt_engine/geometry/HittableList.cuh    HittableList::ClosestIntersection
//     main/src/rt_engine/geometry/bvh_node.cuh        bvh_node::ClosestIntersection
//     main/src/rt_engine/geometry/BVH.cuh             BVH::Node layout (the traversal itself, BVH.cu, does not build: <format>, <<<>>>)
//     main/src/rt_engine/shaders/texture.cuh, cu_Textures.cuh   solid_texture, checker_texture::value
// with plain g++ -std=c++20.  <cuda_runtime.h> is NVIDIA's real header as shipped inside this image's triton wheel
// (triton/backends/nvidia/include) — no stand-in headers are written; g++ ignores the __host__/__device__ attributes.
// Still unbuildable here, genuinely: SphereHittable.cuh / BVH.cu (cuError.h includes <format>, cuda_utils.cuh launches a
// kernel with <<<>>>), materials and cameras (cuRandom.cuh needs curand_kernel.h, which the image does not have).
//
// Leaves of the aggregate probes are a probe-defined Hittable (ProbeSphere) because the reference's SphereHittable does not
// build; its `t` restates _sphere_closest_intersection (SphereHittable.cuh:15-33) and the `t >= rec.distance` reject of
// SphereHittable.cu:58.  What these fixtures pin is the aggregates' own code: the bounds pre-test against rec.distance,
// the visiting order, `hit |=`, and which leaves are reached.
//
// Usage: oracle/_ref/path_probe <out_dir>     (oracle/gen_golden.py)
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "rt_engine/ray_data.cuh"
#include "rt_engine/geometry/aabb.cuh"
#include "rt_engine/geometry/hittable.cuh"
#include "rt_engine/geometry/HittableList.cuh"
#include "rt_engine/geometry/bvh_node.cuh"
#include "rt_engine/geometry/BVH.cuh"
#include "rt_engine/shaders/texture.cuh"
#include "rt_engine/shaders/cu_Textures.cuh"

static uint64_t g_state = 0x2024ull;
static uint32_t next_u32() {  // splitmix64
    uint64_t z = (g_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)((z ^ (z >> 31)) >> 32);
}
static float uni() { return (float)(next_u32() >> 8) * (1.0f / 16777216.0f); }
static float sym(float s) { return (uni() * 2.0f - 1.0f) * s; }
static float special(uint32_t k) {
    const float inf = std::numeric_limits<float>::infinity();
    const float nan = std::numeric_limits<float>::quiet_NaN();
    const float tab[] = {0.0f, -0.0f, inf, -inf, nan, 1e-38f, 1e-42f, -1e-42f, 3.402823466e+38F, -3.402823466e+38F, 1e20f, -1e20f, 1e-20f};
    return tab[k % (sizeof(tab) / sizeof(tab[0]))];
}
static glm::vec3 v3(float s) { return glm::vec3(sym(s), sym(s), sym(s)); }

template <typename T> static void write_file(const std::string& path, const std::vector<T>& d) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) { std::perror(path.c_str()); std::exit(1); }
    std::fwrite(d.data(), sizeof(T), d.size(), f);
    std::fclose(f);
    std::printf("%-40s %zu values\n", path.c_str(), d.size());
}
static void put(std::vector<float>& v, const glm::vec3& a) { v.push_back(a.x); v.push_back(a.y); v.push_back(a.z); }

// ---------------------------------------------------------------------------------------------------------------------
// G1: aabb::intersects (aabb.cuh:30-44)
// ---------------------------------------------------------------------------------------------------------------------
static void gen_aabb(const std::string& dir) {
    const int N = 4096;
    std::vector<float> in, out;
    for (int k = 0; k < N; k++) {
        glm::vec3 a = v3(10.0f), b = v3(10.0f);
        glm::vec3 bmin = glm::min(a, b), bmax = glm::max(a, b);
        const uint32_t kind = k % 16;
        if (kind == 1) { bmin = a; bmax = b; }                                  // possibly inverted per axis
        if (kind == 2) { bmax = bmin; }                                         // degenerate: a point
        if (kind == 3) { bmax[next_u32() % 3] = bmin[next_u32() % 3]; }         // flat / mixed
        if (kind == 4) { aabb e; bmin = e.getMin(); bmax = e.getMax(); }        // the empty box aabb() = (1e9, -1e9), aabb.cuh:17
        glm::vec3 centre = (bmin + bmax) * 0.5f, half = (bmax - bmin) * 0.5f;
        glm::vec3 o = v3(25.0f);
        glm::vec3 target = centre + glm::vec3(sym(1.6f) * half.x, sym(1.6f) * half.y, sym(1.6f) * half.z);   // ~half the rays hit
        glm::vec3 d = target - o;
        if (kind == 5) o = centre + glm::vec3(sym(0.9f) * half.x, sym(0.9f) * half.y, sym(0.9f) * half.z);   // origin inside
        if (kind == 6) d[next_u32() % 3] = 0.0f;                                // axis-parallel slab: (b - o) / 0 = +-inf or NaN
        if (kind == 7) { d[next_u32() % 3] = -0.0f; d[next_u32() % 3] = 0.0f; }
        if (kind == 8) { int ax = next_u32() % 3; d[ax] = 0.0f; o[ax] = (next_u32() & 1) ? bmin[ax] : bmax[ax]; }   // 0 / 0
        if (kind == 9) d[next_u32() % 3] = special(next_u32());
        if (kind == 10) o[next_u32() % 3] = special(next_u32());
        if (kind == 11) d = d * std::ldexp(1.0f, (int)(next_u32() % 60) - 30);  // un-normalised directions of any length
        if (kind == 12) d = -d;                                                 // box behind the origin
        float maxd = 3.402823466e+38F;                                          // _MISS_DIST for a fresh payload
        if (kind == 13) maxd = uni() * 3.0f;                                    // rec.distance already small: culling `tmin < ray_max_dist`
        if (kind == 14) maxd = (next_u32() & 1) ? 0.0f : -1.0f;
        if (kind == 15) bmin[next_u32() % 3] = special(next_u32());
        put(in, bmin); put(in, bmax); put(in, o); put(in, d); in.push_back(maxd);
        float dist = 0.0f;   // `dist` is written only on a hit (aabb.cuh:42); probes initialise it to 0
        bool hit = aabb(bmin, bmax).intersects(Ray(o, d), maxd, dist);
        if (kind == 13 && hit && (k & 16)) {   // exactly at the limit: tmin < ray_max_dist must be false for maxd == tmin
            in.back() = dist; float d2 = 0.0f; hit = aabb(bmin, bmax).intersects(Ray(o, d), dist, d2); dist = d2;
        }
        out.push_back(hit ? 1.0f : 0.0f); out.push_back(dist);
    }
    write_file(dir + "/ref_aabb_in.f32", in);
    write_file(dir + "/ref_aabb_out.f32", out);
}

// aabb helpers used by the BVH builders: longest_axis (aabb.cuh:46-53), surface_area (:55-64), centeroid (:66-68),
// union ctor (:19), operator+= (:24), box_{x,y,z}_compare (:78-88)
static void gen_aabb_misc(const std::string& dir) {
    const int N = 1024;
    std::vector<float> in, out;
    for (int k = 0; k < N; k++) {
        glm::vec3 a0 = v3(10.0f), a1 = v3(10.0f), b0 = v3(10.0f), b1 = v3(10.0f);
        glm::vec3 amin = glm::min(a0, a1), amax = glm::max(a0, a1), bmin = glm::min(b0, b1), bmax = glm::max(b0, b1);
        if (k % 8 == 1) { amin = a0; amax = a1; }                       // inverted on some axis: surface_area returns 0
        if (k % 8 == 2) { aabb e; amin = e.getMin(); amax = e.getMax(); }   // union with the empty box is the other box
        if (k % 8 == 3) { int ax = next_u32() % 3, bx = (ax + 1) % 3; amax[bx] = amin[bx] + (amax[ax] - amin[ax]); }   // equal spans: tie rules of longest_axis
        if (k % 8 == 4) { amax = amin + glm::vec3(1.0f); }               // cube: all spans equal
        if (k % 8 == 5) { bmin[next_u32() % 3] = amin[next_u32() % 3]; }
        put(in, amin); put(in, amax); put(in, bmin); put(in, bmax);
        aabb A(amin, amax), B(bmin, bmax);
        out.push_back((float)A.longest_axis());
        out.push_back(A.surface_area());
        put(out, A.centeroid());
        aabb U(A, B);
        put(out, U.getMin()); put(out, U.getMax());
        aabb P = A; P += B;
        put(out, P.getMin()); put(out, P.getMax());
        out.push_back(box_x_compare(A, B) ? 1.0f : 0.0f);
        out.push_back(box_y_compare(A, B) ? 1.0f : 0.0f);
        out.push_back(box_z_compare(A, B) ? 1.0f : 0.0f);
    }
    write_file(dir + "/ref_aabbmisc_in.f32", in);
    write_file(dir + "/ref_aabbmisc_out.f32", out);
}

// ---------------------------------------------------------------------------------------------------------------------
// aggregates: HittableList (HittableList.cuh:21-34) and bvh_node (bvh_node.cuh:19-24) over probe-defined leaves
// ---------------------------------------------------------------------------------------------------------------------
static std::vector<int>* g_visits = nullptr;

struct ProbeSphere : public Hittable {
    glm::vec3 c; float r; int id;
    ProbeSphere(glm::vec3 c, float r, int id) : c(c), r(r), id(id) {}
    // restated leaf (see the header of this file): _sphere_closest_intersection + SphereHittable::ClosestIntersection's reject
    virtual bool ClosestIntersection(const Ray& ray, RayPayload& rec) const override {
        g_visits->push_back(id);
        glm::vec3 oc = ray.o - c;
        float a = glm::dot(ray.d, ray.d);
        float hb = glm::dot(ray.d, oc);
        float cc = glm::dot(oc, oc) - r * r;
        float d = hb * hb - a * cc;
        float t = _MISS_DIST;
        if (!(d <= 0)) {
            d = sqrtf(d);
            t = (-hb - d) / a;
            if (t < 0.0f) { t = (-hb + d) / a; if (t < 0.0f) t = _MISS_DIST; }
        }
        if (t >= rec.distance) return false;
        rec.distance = t;
        rec.payload.payload[0] = id;
        return true;
    }
};

static aabb sphere_box(const ProbeSphere& s) { return aabb(s.c - glm::vec3(s.r), s.c + glm::vec3(s.r)); }   // getSphereBounds, SphereHittable.cu:52-54

static void gen_aggregates(const std::string& dir) {
    const int S = 64, NS = 8, R = 64;
    std::vector<float> spheres, node_boxes, list_bounds, rays, list_out, tree_out;
    std::vector<int32_t> refs;   // per scenario: 7 x (left, right) in rt_bvh_node / RT_WORLD_NODE_TREE convention, then the root reference
    for (int s = 0; s < S; s++) {
        std::vector<ProbeSphere> sp;
        for (int i = 0; i < NS; i++) sp.emplace_back(v3(4.0f), 0.3f + uni() * 1.5f, i);
        if (s % 4 == 1) { sp[5].c = sp[2].c; sp[5].r = sp[2].r; }           // identical spheres: the first one visited keeps the hit
        if (s % 4 == 2) { sp[1].c = sp[6].c; sp[1].r = sp[6].r; sp[7].c = sp[0].c; sp[7].r = sp[0].r; }
        for (auto& q : sp) { put(spheres, q.c); spheres.push_back(q.r); }
        // --- HittableList: bounds = union of the sphere boxes, in some scenarios deliberately too small (the pre-test then hides spheres)
        aabb lb;
        for (auto& q : sp) lb += sphere_box(q);
        if (s % 8 == 3) lb = aabb(lb.getMin() * 0.5f, lb.getMax() * 0.5f);
        if (s % 8 == 7) lb = aabb(lb.getMin() + glm::vec3(1.5f, 0.0f, 0.0f), lb.getMax());
        put(list_bounds, lb.getMin()); put(list_bounds, lb.getMax());
        std::vector<const Hittable*> ptrs;
        for (auto& q : sp) ptrs.push_back(&q);
        HittableList list(ptrs.data(), NS, lb);
        // --- bvh_node tree: random topology, node bounds = union of the children (sometimes shrunk: the box test gates a subtree)
        struct Item { const Hittable* h; aabb b; int32_t ref; };
        std::vector<Item> items;
        for (int i = 0; i < NS; i++) items.push_back({&sp[i], sphere_box(sp[i]), -(i + 1)});
        std::vector<bvh_node*> nodes;
        while (items.size() > 1) {
            size_t i = next_u32() % items.size();
            Item a = items[i]; items.erase(items.begin() + i);
            size_t j = next_u32() % items.size();
            Item b = items[j]; items.erase(items.begin() + j);
            aabb nb(a.b, b.b);
            if (s % 8 == 5 && (next_u32() % 3) == 0) { glm::vec3 c = nb.centeroid(); nb = aabb(c + (nb.getMin() - c) * 0.6f, c + (nb.getMax() - c) * 0.6f); }
            bvh_node* n = new bvh_node(a.h, b.h, nb);
            put(node_boxes, nb.getMin()); put(node_boxes, nb.getMax());
            refs.push_back(a.ref); refs.push_back(b.ref);
            items.push_back({n, nb, (int32_t)nodes.size()});
            nodes.push_back(n);
        }
        refs.push_back(items[0].ref);
        const Hittable* tree = items[0].h;
        // --- rays
        for (int k = 0; k < R; k++) {
            glm::vec3 o = v3(9.0f);
            if (k % 8 == 1) o = sp[next_u32() % NS].c + v3(0.2f);             // starts inside a sphere
            glm::vec3 d = (sp[next_u32() % NS].c + v3(0.8f)) - o;
            if (k % 8 == 2) d[next_u32() % 3] = 0.0f;
            if (k % 8 == 3) d = d * std::ldexp(1.0f, (int)(next_u32() % 16) - 8);
            if (k % 8 == 4) d = v3(1.0f);                                    // mostly misses
            put(rays, o); put(rays, d); rays.push_back(0.0f);
            Ray ray(o, d);
            for (int which = 0; which < 2; which++) {
                std::vector<int> visits;
                g_visits = &visits;
                RayPayload rec;   // distance = _MISS_DIST (ray_data.cuh:40): what sample_world starts every trace with (Renderer.cu:147)
                rec.payload.payload[0] = -1;
                bool hit = which == 0 ? list.ClosestIntersection(ray, rec) : tree->ClosestIntersection(ray, rec);
                std::vector<float>& out = which == 0 ? list_out : tree_out;
                out.push_back(hit ? 1.0f : 0.0f);
                out.push_back(rec.distance);
                out.push_back((float)rec.payload.payload[0]);
                out.push_back((float)visits.size());
                for (int v = 0; v < NS; v++) out.push_back(v < (int)visits.size() ? (float)visits[v] : -1.0f);   // leaf visiting order
            }
        }
        for (auto* n : nodes) delete n;
    }
    write_file(dir + "/ref_agg_spheres.f32", spheres);
    write_file(dir + "/ref_agg_nodeboxes.f32", node_boxes);
    write_file(dir + "/ref_agg_refs.i32", refs);
    write_file(dir + "/ref_agg_listbounds.f32", list_bounds);
    write_file(dir + "/ref_agg_rays.f32", rays);
    write_file(dir + "/ref_agg_list_out.f32", list_out);
    write_file(dir + "/ref_agg_tree_out.f32", tree_out);
}

// ---------------------------------------------------------------------------------------------------------------------
// checker_texture::value (cu_Textures.cuh:31-39) over two solid_textures; Ray::at (ray_data.cuh:14); isBackfacing (:44-46)
// ---------------------------------------------------------------------------------------------------------------------
static void gen_checker(const std::string& dir) {
    const int N = 2048;
    std::vector<float> in, out;
    for (int k = 0; k < N; k++) {
        glm::vec3 even(uni(), uni(), uni()), odd(uni(), uni(), uni());
        const float scales[] = {0.32f, 1.0f, 3.0f, 0.5f, 10.0f, 0.01f};
        float scale = scales[next_u32() % 6];
        glm::vec3 pos = v3((k % 3 == 0) ? 2.0f : 40.0f);
        if (k % 8 == 1) pos = glm::vec3((float)((int)(next_u32() % 9) - 4), (float)((int)(next_u32() % 9) - 4), (float)((int)(next_u32() % 9) - 4)) * scale;   // on cell boundaries
        if (k % 8 == 2) pos[next_u32() % 3] = -uni() * 0.999f * scale;        // (-1, 0) truncates to 0, not -1: the book's floor would differ
        if (k % 8 == 3) pos[next_u32() % 3] = -0.0f;
        solid_texture te(even), to(odd);
        checker_texture tex(&te, &to, scale);
        put(in, even); put(in, odd); in.push_back(scale); put(in, pos);
        put(out, tex.value(glm::vec2(0.0f, 0.0f), pos));
    }
    write_file(dir + "/ref_checker_in.f32", in);
    write_file(dir + "/ref_checker_out.f32", out);
}

static void gen_ray(const std::string& dir) {
    const int N = 1024;
    std::vector<float> in, out;
    for (int k = 0; k < N; k++) {
        glm::vec3 o = v3(20.0f), d = v3(2.0f), n = v3(1.0f);
        float t = (k & 1) ? uni() * 50.0f : sym(1e3f);
        if (k % 16 == 3) d[next_u32() % 3] = special(next_u32());
        if (k % 16 == 5) n = glm::vec3(d.y, -d.x, 0.0f);   // perpendicular: dot == 0 is front-facing
        put(in, o); put(in, d); in.push_back(t); put(in, n);
        put(out, Ray(o, d).at(t));
        out.push_back(isBackfacing(Ray(o, d), n) ? 1.0f : 0.0f);
    }
    write_file(dir + "/ref_ray_in.f32", in);
    write_file(dir + "/ref_ray_out.f32", out);
}

// record layouts of the reference: what the flat records of include/rt06.h mirror
static void gen_layout(const std::string& dir) {
    FILE* f = std::fopen((dir + "/ref_layout.json").c_str(), "w");
    if (!f) { std::perror("ref_layout.json"); std::exit(1); }
    std::fprintf(f,
                 "{\"sizeof_Ray\": %zu, \"sizeof_RayPayload\": %zu, \"sizeof_aabb\": %zu, \"sizeof_BVH_Node\": %zu, "
                 "\"offsetof_BVH_Node_left_child_idx\": %zu, \"offsetof_BVH_Node_right_child_hittable_idx\": %zu, "
                 "\"offsetof_Ray_d\": %zu, \"offsetof_Ray_time\": %zu, \"offsetof_RayPayload_distance\": %zu, "
                 "\"MISS_DIST_bits\": %u, \"IS_LEAF_CODE\": %d}\n",
                 sizeof(Ray), sizeof(RayPayload), sizeof(aabb), sizeof(BVH::Node), offsetof(BVH::Node, left_child_idx),
                 offsetof(BVH::Node, right_child_hittable_idx), offsetof(Ray, d), offsetof(Ray, time), offsetof(RayPayload, distance),
                 [] { float m = _MISS_DIST; uint32_t u; std::memcpy(&u, &m, 4); return u; }(), _IS_LEAF_CODE);
    std::fclose(f);
    std::printf("%s/ref_layout.json\n", dir.c_str());
}

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s <out_dir>\n", argv[0]); return 2; }
    const std::string dir = argv[1];
    gen_aabb(dir);
    gen_aabb_misc(dir);
    gen_aggregates(dir);
    gen_checker(dir);
    gen_ray(dir);
    gen_layout(dir);
    return 0;
}

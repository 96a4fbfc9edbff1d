// rt_host.cpp — host side of the C ABI: cameras, scene vocabulary, BVH builders, flattening.
//
// The reference builds its world out of ~1.5K single-object device allocations, each constructed by
// a <<<1,1>>> placement-new kernel so that vtables are device-valid (utilities/cuda_utilities/
// cuda_utils.cuh:9-23; SphereHittable.cuh:134-154).  Here the same vocabulary produces three flat
// host arrays (nodes 32 B, primitives 32 B, materials 32 B) that one hipMemcpy each puts in HBM.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "rt06.h"
#include "rt_internal.hpp"
#include "rt_math.hpp"

// ---------------------------------------------------------------------------------------------
// error reporting
// ---------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

int rt_fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}
extern "C" const char* rt_last_error(void) { return g_last_error.c_str(); }
#ifndef RT06_SRC_SHA256
#define RT06_SRC_SHA256 "unstamped"
#endif
extern "C" const char* rt_version(void) { return "rt06-amd 0.3 (gfx950) src " RT06_SRC_SHA256; }
extern "C" const char* rt_source_hash(void) { return RT06_SRC_SHA256; }

// ---------------------------------------------------------------------------------------------
// cameras — rt_engine/shaders/cu_Cameras.cuh ctors (:16-25, :40-52, :73-85)
// ---------------------------------------------------------------------------------------------
static void camera_basis(const float lookfrom[3], const float lookat[3], const float up[3], float vfov, float aspect,
                         bool prescale, rt_camera* c) {
    float theta = radians(vfov);
    float vh = tanf(theta * 0.5f);
    float vw = vh * aspect;
    f3 w = normalize(ld3(lookat) - ld3(lookfrom));
    f3 u = normalize(cross(ld3(up), w));
    if (prescale) u = u * vw;
    f3 v = normalize(cross(w, u));
    if (prescale) v = v * vh;
    std::memset(c, 0, sizeof(*c));
    st3(c->o, ld3(lookfrom)); st3(c->u, u); st3(c->v, v); st3(c->w, w);
    c->viewport_width = vw; c->viewport_height = vh;
    c->t0 = 0.0f; c->t1 = 1.0f;
}
extern "C" int rt_camera_pinhole(const float lookfrom[3], const float lookat[3], const float up[3], float vfov,
                                 float aspect, rt_camera* out) {
    if (!lookfrom || !lookat || !up || !out) return rt_fail(RT_ERR_INVALID, "rt_camera_pinhole: null argument");
    camera_basis(lookfrom, lookat, up, vfov, aspect, true, out);
    out->type = RT_CAM_PINHOLE;
    return RT_OK;
}
extern "C" int rt_camera_defocus(const float lookfrom[3], const float lookat[3], const float up[3], float vfov,
                                 float aspect, float aperture, float focus_dist, rt_camera* out) {
    if (!lookfrom || !lookat || !up || !out) return rt_fail(RT_ERR_INVALID, "rt_camera_defocus: null argument");
    camera_basis(lookfrom, lookat, up, vfov, aspect, false, out);
    out->type = RT_CAM_DEFOCUS;
    out->lens_radius = aperture * 0.5f;
    out->focus_dist = focus_dist;
    return RT_OK;
}
extern "C" int rt_camera_motion(const float lookfrom[3], const float lookat[3], const float up[3], float vfov,
                                float aspect, float time0, float time1, rt_camera* out) {
    if (!lookfrom || !lookat || !up || !out) return rt_fail(RT_ERR_INVALID, "rt_camera_motion: null argument");
    camera_basis(lookfrom, lookat, up, vfov, aspect, true, out);
    out->type = RT_CAM_MOTION;
    out->t0 = time0; out->t1 = time1;
    return RT_OK;
}

// ---------------------------------------------------------------------------------------------
// aabb — rt_engine/geometry/aabb.cuh (host-side members used by the builders)
// ---------------------------------------------------------------------------------------------
struct Box {
    f3 mn, mx;
};
static Box box_empty() { return Box{mk3(1e9f), mk3(-1e9f)}; }                                  // aabb.cuh:17
static Box box_union(const Box& a, const Box& b) { return Box{glm_min(a.mn, b.mn), glm_max(a.mx, b.mx)}; }  // :19,:24
static int box_longest_axis(const Box& b) {                                                    // :46-53
    f3 s = b.mx - b.mn;
    s = mk3(fabsf(s.x), fabsf(s.y), fabsf(s.z));
    if (s.x > s.y) return s.x > s.z ? 0 : 2;
    return s.y > s.z ? 1 : 2;
}
static float box_surface_area(const Box& b) {                                                  // :55-64
    f3 s = b.mx - b.mn;
    if (s.x < 0 || s.y < 0 || s.z < 0) return 0.0f;
    float cost = 0.0f;
    cost += s.x * s.y; cost += s.x * s.z; cost += s.y * s.z;
    return 2.0f * cost;
}
static f3 box_centroid(const Box& b) { return (b.mx + b.mn) * 0.5f; }                          // :66-68
static float axis_of(f3 a, int ax) { return ax == 0 ? a.x : (ax == 1 ? a.y : a.z); }

// getSphereBounds / getMovingSphereBounds, SphereHittable.cu:52-54, 85-89
// Host probe: the aabb helpers the builders are made of (aabb.cuh:19,24,46-68,78-88), over arrays — checked against fixtures
// generated from the reference's own aabb.cuh (tests/golden/ref_aabbmisc_*).  boxes n*12 (a.min a.max b.min b.max) ->
// out n*20 [longest axis, surface area, centroid 3, union min 3 max 3, a += b min 3 max 3, a.min < b.min per axis 3]
extern "C" int rt_probe_aabb_misc(size_t n, const float* boxes, float* out) {
    if (!boxes || !out) return rt_fail(RT_ERR_INVALID, "rt_probe_aabb_misc: null argument");
    for (size_t i = 0; i < n; i++) {
        const Box a{ld3(boxes + 12 * i), ld3(boxes + 12 * i + 3)}, b{ld3(boxes + 12 * i + 6), ld3(boxes + 12 * i + 9)};
        float* o = out + 20 * i;
        o[0] = (float)box_longest_axis(a);
        o[1] = box_surface_area(a);
        st3(o + 2, box_centroid(a));
        const Box u = box_union(a, b);
        st3(o + 5, u.mn); st3(o + 8, u.mx);
        Box acc = a;
        acc = box_union(acc, b);
        st3(o + 11, acc.mn); st3(o + 14, acc.mx);
        o[17] = a.mn.x < b.mn.x ? 1.0f : 0.0f; o[18] = a.mn.y < b.mn.y ? 1.0f : 0.0f; o[19] = a.mn.z < b.mn.z ? 1.0f : 0.0f;
    }
    return RT_OK;
}

static Box prim_bounds(const rt_prim& p) {
    f3 r = mk3(p.radius);
    Box b0{ld3(p.c0) - r, ld3(p.c0) + r};
    if (!(p.mat & RT_PRIM_MOVING)) return b0;
    Box b1{ld3(p.c1) - r, ld3(p.c1) + r};
    return box_union(b0, b1);
}

// ---------------------------------------------------------------------------------------------
// rt_scene
// ---------------------------------------------------------------------------------------------
struct rt_scene {
    std::vector<rt_prim> prims;
    std::vector<rt_quad> quads;
    uint32_t background = 0;
    float background_color[3] = {0.0f, 0.0f, 0.0f};
    uint32_t traversal = RT_TRAVERSAL_STACK;
    std::vector<rt_material> mats;
    std::vector<rt_perlin> perlin;        // 0 or 1 table set (RT_MAT_LAMBERTIAN_NOISE)
    std::vector<uint8_t> image;           // RGB8 (RT_MAT_LAMBERTIAN_IMAGE)
    uint32_t image_w = 0, image_h = 0;
    std::vector<rt_bvh_node> nodes;       // world nodes (BVH or bvh_node tree)
    std::vector<rt_bvh_node> tree_nodes;  // bvh_node objects created by rt_scene_add_bvh_node
    uint32_t kind = RT_WORLD_LIST;
    int32_t root = 0;
    uint32_t max_stack = 0;
    Box bounds = box_empty();
    bool world_set = false;
};

extern "C" int rt_scene_create(rt_scene** out) {
    if (!out) return rt_fail(RT_ERR_INVALID, "rt_scene_create: null out");
    *out = new rt_scene();
    return RT_OK;
}
extern "C" void rt_scene_destroy(rt_scene* s) { delete s; }

extern "C" int rt_scene_add_material(rt_scene* s, uint32_t type, const float albedo[3], float param,
                                     const float albedo2[3], int32_t* out_id) {
    if (!s || !albedo) return rt_fail(RT_ERR_INVALID, "rt_scene_add_material: null argument");
    if (type > RT_MAT_LAMBERTIAN_IMAGE) return rt_fail(RT_ERR_INVALID, "rt_scene_add_material: unknown material type %u", type);
    if (type == RT_MAT_ISOTROPIC && !(param > 0.0f)) return rt_fail(RT_ERR_INVALID, "rt_scene_add_material: a constant medium needs a density > 0");
    rt_material m;
    std::memset(&m, 0, sizeof(m));
    st3(m.albedo, ld3(albedo));
    m.param = param;
    if (albedo2) st3(m.albedo2, ld3(albedo2));
    m.type = type;
    s->mats.push_back(m);
    if (out_id) *out_id = (int32_t)s->mats.size() - 1;
    return RT_OK;
}

static int add_prim(rt_scene* s, const float c0[3], const float c1[3], float radius, int32_t mat, bool moving, int32_t* out_prim) {
    if (!s || !c0 || !c1) return rt_fail(RT_ERR_INVALID, "add sphere: null argument");
    if (mat < 0 || (size_t)mat >= s->mats.size()) return rt_fail(RT_ERR_INVALID, "add sphere: material index %d out of range", mat);
    rt_prim p;
    st3(p.c0, ld3(c0)); st3(p.c1, ld3(c1));
    p.radius = radius;
    p.mat = (uint32_t)mat | (moving ? RT_PRIM_MOVING : 0u);
    s->prims.push_back(p);
    s->world_set = false;
    if (out_prim) *out_prim = (int32_t)s->prims.size() - 1;
    return RT_OK;
}
extern "C" int rt_scene_add_sphere(rt_scene* s, const float center[3], float radius, int32_t mat, int32_t* out_prim) {
    return add_prim(s, center, center, radius, mat, false, out_prim);
}
extern "C" int rt_scene_add_moving_sphere(rt_scene* s, const float c0[3], const float c1[3], float radius, int32_t mat,
                                          int32_t* out_prim) {
    return add_prim(s, c0, c1, radius, mat, true, out_prim);
}
extern "C" int rt_scene_prim_bounds(const rt_scene* s, int32_t prim, float out_min[3], float out_max[3]) {
    if (!s || prim < 0 || (size_t)prim >= s->prims.size()) return rt_fail(RT_ERR_INVALID, "rt_scene_prim_bounds: bad primitive %d", prim);
    Box b = prim_bounds(s->prims[prim]);
    st3(out_min, b.mn); st3(out_max, b.mx);
    return RT_OK;
}

// ---- quads (not in the reference; "Ray Tracing: The Next Week" quad(Q,u,v)) -------------------------
static void quad_finalize(rt_quad& q) {
    f3 n = cross(ld3(q.u), ld3(q.v));
    f3 normal = normalize(n);
    st3(q.normal, normal);
    q.D = dot(normal, ld3(q.Q));
    st3(q.w, n / dot(n, n));
    q.pad0 = q.pad1 = q.pad2 = 0.0f;
}
static Box box_of_points(f3 a, f3 b) { return Box{glm_min(a, b), glm_max(a, b)}; }
// bbox = box(Q, Q+u+v) U box(Q+u, Q+v), every axis padded to at least 0.0001 (the book's aabb::pad_to_minimums)
static Box quad_bounds(const rt_quad& q) {
    f3 Q = ld3(q.Q), u = ld3(q.u), v = ld3(q.v);
    Box b = box_union(box_of_points(Q, Q + u + v), box_of_points(Q + u, Q + v));
    const float delta = 0.0001f;
    if (b.mx.x - b.mn.x < delta) { b.mn.x -= delta / 2; b.mx.x += delta / 2; }
    if (b.mx.y - b.mn.y < delta) { b.mn.y -= delta / 2; b.mx.y += delta / 2; }
    if (b.mx.z - b.mn.z < delta) { b.mn.z -= delta / 2; b.mx.z += delta / 2; }
    return b;
}
extern "C" int rt_scene_add_quad(rt_scene* s, const float Q[3], const float u[3], const float v[3], int32_t mat, int32_t* out_quad) {
    if (!s || !Q || !u || !v) return rt_fail(RT_ERR_INVALID, "rt_scene_add_quad: null argument");
    if (mat < 0 || (size_t)mat >= s->mats.size()) return rt_fail(RT_ERR_INVALID, "rt_scene_add_quad: material index %d out of range", mat);
    rt_quad q;
    std::memset(&q, 0, sizeof(q));
    st3(q.Q, ld3(Q)); st3(q.u, ld3(u)); st3(q.v, ld3(v));
    q.mat = (uint32_t)mat;
    quad_finalize(q);
    s->quads.push_back(q);
    s->world_set = false;
    if (out_quad) *out_quad = (int32_t)s->quads.size() - 1;
    return RT_OK;
}
extern "C" int rt_scene_set_background(rt_scene* s, uint32_t mode, const float color[3]) {
    if (!s) return rt_fail(RT_ERR_INVALID, "rt_scene_set_background: null scene");
    if (mode > 1) return rt_fail(RT_ERR_INVALID, "rt_scene_set_background: unknown mode %u", mode);
    s->background = mode;
    if (color) st3(s->background_color, ld3(color));
    return RT_OK;
}

extern "C" int rt_scene_set_traversal(rt_scene* s, uint32_t mode) {
    if (!s) return rt_fail(RT_ERR_INVALID, "rt_scene_set_traversal: null scene");
    if (mode > RT_TRAVERSAL_WIDE4) return rt_fail(RT_ERR_INVALID, "rt_scene_set_traversal: unknown mode %u", mode);
    s->traversal = mode;
    return RT_OK;
}

// ---- BVH_Handle::Factory (rt_engine/geometry/BVH.cu:156-384) --------------------------------
namespace {
struct Item {
    Box b;
    bool is_quad;
    rt_prim p;
    rt_quad q;
};
struct Builder {
    std::vector<Item> arr;
    std::vector<rt_bvh_node>& nodes;
    explicit Builder(std::vector<rt_bvh_node>& n) : nodes(n) {}

    int32_t push(const Box& b, int32_t l, int32_t r) {
        rt_bvh_node n;
        st3(n.min, b.mn); st3(n.max, b.mx);
        n.left = l; n.right = r;
        nodes.push_back(n);
        return (int32_t)nodes.size() - 1;
    }
    Box partition_bounds(int start, int end) const {  // _get_partition_bounds, BVH.cu:306-312
        Box b = box_empty();
        for (int i = start; i < end; i++) b = box_union(b, arr[i].b);
        return b;
    }
    // The reference sorts with std::sort (unstable) on bounds.min[axis] only (BVH.cu:195-199,
    // aabb.cuh:78-88), so equal keys land in an STL-specific order.  A stable sort pins that order.
    void sort_axis(int start, int end, int axis) {
        std::stable_sort(arr.begin() + start, arr.begin() + end,
                         [axis](const Item& a, const Item& b) { return axis_of(a.b.mn, axis) < axis_of(b.b.mn, axis); });
    }
    int32_t rec1(int start, int end) {  // _build_bvh_rec1, BVH.cu:180-210
        Box bounds = partition_bounds(start, end);
        int axis = box_longest_axis(bounds);
        if (end - start == 1) return push(bounds, -1, start);
        sort_axis(start, end, axis);
        int mid = (start + end) / 2;
        int32_t l = rec1(start, mid);
        int32_t r = rec1(mid, end);
        return push(bounds, l, r);
    }
    void find_optimal_split(int start, int end, const Box& bounds, int& best_axis, float& best_split) const {  // BVH.cu:241-279
        const int split_points = 16;
        float best_cost = RT_MISS_DIST;
        best_axis = 0; best_split = 0.0f;
        for (int axis = 0; axis < 3; axis++)
            for (int split = 0; split < split_points; split++) {
                float pos = ((float)split + 1.0f) / ((float)split_points + 1.0f);
                pos = mix(axis_of(bounds.mn, axis), axis_of(bounds.mx, axis), pos);
                Box lb = box_empty(), rb = box_empty();
                int lc = 0, rc = 0;
                for (int i = start; i < end; i++) {
                    const Box& b = arr[i].b;
                    if (axis_of(box_centroid(b), axis) < pos) { lb = box_union(lb, b); lc++; }
                    else { rb = box_union(rb, b); rc++; }
                }
                float cost = box_surface_area(lb) * (float)lc + box_surface_area(rb) * (float)rc;
                if (cost < best_cost) { best_cost = cost; best_axis = axis; best_split = pos; }
            }
    }
    int partition_by_split(int start, int end, int axis, float split_pos) {  // BVH.cu:281-304
        int i = start, j = end;
        while (i < j) {
            if (axis_of(box_centroid(arr[i].b), axis) < split_pos) i++;
            else std::swap(arr[i], arr[--j]);
        }
        return i;
    }
    // _build_bvh_rec2, BVH.cu:212-239.  When the chosen plane leaves one side empty the reference
    // recurses on an empty range forever; this falls back to rec1's median split for that range.
    int32_t rec2(int start, int end) {
        Box bounds = partition_bounds(start, end);
        if (end - start == 1) return push(bounds, -1, start);
        int axis; float split;
        find_optimal_split(start, end, bounds, axis, split);
        int mid = partition_by_split(start, end, axis, split);
        if (mid == start || mid == end) {
            sort_axis(start, end, box_longest_axis(bounds));
            mid = (start + end) / 2;
        }
        int32_t l = rec2(start, mid);
        int32_t r = rec2(mid, end);
        return push(bounds, l, r);
    }
    // BuildBVH_BottomUp + _find_optimal_merge + _merge_nodes, BVH.cu:315-384
    int32_t bottom_up() {
        struct BN { Box b; int count; int32_t idx; };
        std::vector<BN> bn;
        for (size_t i = 0; i < arr.size(); i++) bn.push_back(BN{arr[i].b, 1, push(arr[i].b, -1, (int32_t)i)});
        while (bn.size() > 1) {
            float best_cost = RT_MISS_DIST;
            size_t ba = 0, bb = 1;
            for (size_t a = 0; a < bn.size(); a++)
                for (size_t b = a + 1; b < bn.size(); b++) {
                    Box nb = box_union(bn[a].b, bn[b].b);
                    float cost = box_surface_area(nb) * (float)(bn[a].count + bn[b].count);
                    if (cost < best_cost) { ba = a; bb = b; best_cost = cost; }
                }
            BN merged{box_union(bn[ba].b, bn[bb].b), bn[ba].count + bn[bb].count, 0};
            merged.idx = push(merged.b, bn[ba].idx, bn[bb].idx);
            bn.erase(bn.begin() + bb);
            bn.erase(bn.begin() + ba);
            bn.push_back(merged);
        }
        return bn[0].idx;
    }
};

uint32_t leaf_depth(const std::vector<rt_bvh_node>& nodes, int32_t idx) {
    if (nodes[idx].left == -1) return 0;
    return 1 + std::max(leaf_depth(nodes, nodes[idx].left), leaf_depth(nodes, nodes[idx].right));
}
}  // namespace

static int build_bvh(rt_scene* s, int builder) {
    if (!s) return rt_fail(RT_ERR_INVALID, "build_bvh: null scene");
    if (s->prims.empty() && s->quads.empty()) return rt_fail(RT_ERR_INVALID, "build_bvh: scene has no primitives");
    s->nodes.clear();
    Builder B(s->nodes);
    B.arr.reserve(s->prims.size() + s->quads.size());
    for (const rt_prim& p : s->prims) B.arr.push_back(Item{prim_bounds(p), false, p, rt_quad{}});
    for (const rt_quad& q : s->quads) B.arr.push_back(Item{quad_bounds(q), true, rt_prim{}, q});
    int32_t root;
    if (builder == 0) root = B.rec1(0, (int)B.arr.size());
    else if (builder == 1) root = B.rec2(0, (int)B.arr.size());
    else root = B.bottom_up();
    // hittables[] = arr order (BVH.cu:174-177), kept per kind: spheres first, then quads; leaf indices are remapped
    {
        std::vector<int32_t> unified(B.arr.size());
        const size_t ns = s->prims.size();
        size_t si = 0, qi = 0;
        for (size_t i = 0; i < B.arr.size(); i++) {
            if (B.arr[i].is_quad) { s->quads[qi] = B.arr[i].q; unified[i] = (int32_t)(ns + qi); qi++; }
            else { s->prims[si] = B.arr[i].p; unified[i] = (int32_t)si; si++; }
        }
        for (rt_bvh_node& n : s->nodes)
            if (n.left == -1) n.right = unified[n.right];
    }
    s->kind = RT_WORLD_BVH;
    s->root = root;
    s->max_stack = leaf_depth(s->nodes, root) + 1;
    s->bounds = Box{ld3(s->nodes[root].min), ld3(s->nodes[root].max)};
    s->world_set = true;
    // The reference's traversal stack holds 32 entries and is never checked (BVH.cu:17,27-35).
    if (s->max_stack > RT_MAX_STACK)
        return rt_fail(RT_ERR_STACK, "BVH needs a traversal stack of %u entries; the limit is %d (BVH.cu:17)", s->max_stack, RT_MAX_STACK);
    return RT_OK;
}
extern "C" int rt_scene_build_bvh_topdown(rt_scene* s) { return build_bvh(s, 0); }
extern "C" int rt_scene_build_bvh_sah(rt_scene* s) { return build_bvh(s, 1); }
extern "C" int rt_scene_build_bvh_bottomup(rt_scene* s) { return build_bvh(s, 2); }

extern "C" int rt_scene_set_world_list(rt_scene* s) {
    if (!s) return rt_fail(RT_ERR_INVALID, "rt_scene_set_world_list: null scene");
    if (s->prims.empty() && s->quads.empty()) return rt_fail(RT_ERR_INVALID, "rt_scene_set_world_list: scene has no primitives");
    s->nodes.clear();
    s->kind = RT_WORLD_LIST;
    s->root = 0;
    s->max_stack = 0;
    Box b = box_empty();
    for (const rt_prim& p : s->prims) b = box_union(b, prim_bounds(p));  // world_bounds += handle.getBounds(), Scenes.cu:61
    for (const rt_quad& q : s->quads) b = box_union(b, quad_bounds(q));
    s->bounds = b;
    s->world_set = true;
    return RT_OK;
}

static bool ref_valid(const rt_scene* s, int32_t ref) {
    if (ref >= 0) return (size_t)ref < s->tree_nodes.size();
    return (size_t)(-ref - 1) < s->prims.size();
}
static Box ref_bounds(const rt_scene* s, int32_t ref) {
    if (ref >= 0) return Box{ld3(s->tree_nodes[ref].min), ld3(s->tree_nodes[ref].max)};
    return prim_bounds(s->prims[-ref - 1]);
}
extern "C" int rt_scene_add_bvh_node(rt_scene* s, int32_t left_ref, int32_t right_ref, const float bmin[3],
                                     const float bmax[3], int32_t* out_ref) {
    if (!s || !out_ref) return rt_fail(RT_ERR_INVALID, "rt_scene_add_bvh_node: null argument");
    if (!ref_valid(s, left_ref) || !ref_valid(s, right_ref)) return rt_fail(RT_ERR_INVALID, "rt_scene_add_bvh_node: bad child reference");
    Box b = (bmin && bmax) ? Box{ld3(bmin), ld3(bmax)} : box_union(ref_bounds(s, left_ref), ref_bounds(s, right_ref));
    rt_bvh_node n;
    st3(n.min, b.mn); st3(n.max, b.mx);
    n.left = left_ref; n.right = right_ref;
    s->tree_nodes.push_back(n);
    *out_ref = (int32_t)s->tree_nodes.size() - 1;
    return RT_OK;
}
static uint32_t tree_depth(const rt_scene* s, int32_t ref, uint32_t guard) {
    if (ref < 0 || guard > 100000) return 0;
    return 1 + std::max(tree_depth(s, s->tree_nodes[ref].left, guard + 1), tree_depth(s, s->tree_nodes[ref].right, guard + 1));
}
extern "C" int rt_scene_set_world_node_tree(rt_scene* s, int32_t root_ref) {
    if (!s) return rt_fail(RT_ERR_INVALID, "rt_scene_set_world_node_tree: null scene");
    if (!ref_valid(s, root_ref)) return rt_fail(RT_ERR_INVALID, "rt_scene_set_world_node_tree: bad root reference");
    if (!s->quads.empty()) return rt_fail(RT_ERR_INVALID, "rt_scene_set_world_node_tree: bvh_node trees take spheres only");
    s->nodes = s->tree_nodes;
    s->kind = RT_WORLD_NODE_TREE;
    s->root = root_ref;
    // iterative twin of the recursion pushes right then left: one pending sibling per level + 1
    s->max_stack = tree_depth(s, root_ref, 0) + 1;
    s->bounds = ref_bounds(s, root_ref);
    s->world_set = true;
    if (s->max_stack > RT_MAX_STACK)
        return rt_fail(RT_ERR_STACK, "bvh_node tree needs a traversal stack of %u entries; the limit is %d", s->max_stack, RT_MAX_STACK);
    return RT_OK;
}

extern "C" int rt_scene_get_flat(const rt_scene* s, rt_world_flat* out) {
    if (!s || !out) return rt_fail(RT_ERR_INVALID, "rt_scene_get_flat: null argument");
    if (!s->world_set) return rt_fail(RT_ERR_INVALID, "rt_scene_get_flat: no world built (call a build_bvh / set_world function first)");
    std::memset(out, 0, sizeof(*out));
    out->kind = s->kind;
    out->root = s->root;
    out->n_nodes = (uint32_t)s->nodes.size();
    out->n_prims = (uint32_t)s->prims.size();
    out->n_materials = (uint32_t)s->mats.size();
    out->max_stack = s->max_stack;
    st3(out->bounds_min, s->bounds.mn); st3(out->bounds_max, s->bounds.mx);
    out->nodes = s->nodes.empty() ? nullptr : s->nodes.data();
    out->prims = s->prims.empty() ? nullptr : s->prims.data();
    out->materials = s->mats.data();
    out->quads = s->quads.empty() ? nullptr : s->quads.data();
    out->n_quads = (uint32_t)s->quads.size();
    out->background = s->background;
    st3(out->background_color, ld3(s->background_color));
    out->perlin = s->perlin.empty() ? nullptr : s->perlin.data();
    out->image = s->image.empty() ? nullptr : s->image.data();
    out->image_width = s->image_w;
    out->image_height = s->image_h;
    out->traversal = s->kind == RT_WORLD_BVH ? s->traversal : (uint32_t)RT_TRAVERSAL_STACK;
    return RT_OK;
}

// perlin::perlin() ("The Next Week"): randvec[i] = unit_vector(vec3::random(-1, 1)); perm = identity shuffled by
// `for i = n-1 .. 1: swap(p[i], p[random_int(0, i)])`, three times.  Uniforms: the build's sequential host stream, id 0x9E81.
extern "C" int rt_scene_set_perlin(rt_scene* s, uint64_t seed) {
    if (!s) return rt_fail(RT_ERR_INVALID, "rt_scene_set_perlin: null scene");
    Rng g;
    g.init(seed, 0u, 0u, 0x9E81u);
    s->perlin.assign(1, rt_perlin());
    rt_perlin& t = s->perlin[0];
    for (int i = 0; i < 256; i++) {
        f3 v;
        v.x = g.next() * 2.0f - 1.0f;
        v.y = g.next() * 2.0f - 1.0f;
        v.z = g.next() * 2.0f - 1.0f;
        if (near_zero(v)) v = mk3(1.0f, 0.0f, 0.0f);
        st3(t.randvec[i], normalize(v));
    }
    for (int k = 0; k < 3; k++) {
        for (int i = 0; i < 256; i++) t.perm[k][i] = i;
        for (int i = 255; i > 0; i--) {
            int target = (int)(g.next() * (float)(i + 1));   // random_int(0, i); the uniform is in (0, 1]
            if (target > i) target = i;
            int32_t tmp = t.perm[k][i]; t.perm[k][i] = t.perm[k][target]; t.perm[k][target] = tmp;
        }
    }
    return RT_OK;
}

extern "C" int rt_scene_set_image(rt_scene* s, uint32_t width, uint32_t height, const uint8_t* rgb) {
    if (!s || !rgb) return rt_fail(RT_ERR_INVALID, "rt_scene_set_image: null argument");
    if (width == 0 || height == 0 || (uint64_t)width * height > (1ull << 26)) return rt_fail(RT_ERR_INVALID, "rt_scene_set_image: bad size %u x %u", width, height);
    s->image.assign(rgb, rgb + (size_t)width * height * 3);
    s->image_w = width;
    s->image_h = height;
    return RT_OK;
}

extern "C" int rt_host_uniforms(uint64_t seed, uint32_t first, uint32_t n, float* out) {
    if (!out && n) return rt_fail(RT_ERR_INVALID, "rt_host_uniforms: null out");
    Rng g;
    g.init(seed, 0u, 0u, RT_STREAM_SCENE);
    for (uint32_t i = 0; i < first; i++) (void)g.next();  // a sequential stream: skip to `first`
    for (uint32_t i = 0; i < n; i++) out[i] = g.next();
    return RT_OK;
}

// ---- prefab scenes ---------------------------------------------------------------------------
static void add(rt_scene* s, f3 c0, f3 c1, float r, bool moving, uint32_t type, f3 albedo, float param) {
    float a[3], p0[3], p1[3];
    st3(a, albedo); st3(p0, c0); st3(p1, c1);
    int32_t mat = 0;
    rt_scene_add_material(s, type, a, param, nullptr, &mat);
    add_prim(s, p0, p1, r, mat, moving, nullptr);
}

// SceneBook2BVH::Factory::_populate_world (Scenes.cu:219-270) when moving; the disabled SceneBook1
// variant (Scenes.cu:57-115) with static Lambertians otherwise (its centre1 draw is still consumed so
// both scenes share one layout, as google_testing/test.cpp:41-51 does).  Uniforms: the build's host
// stream (RT_STREAM_SCENE) in place of cuHostRND(512, 1984).
static rt_scene* book_scene(uint64_t seed, bool moving) {
    rt_scene* s = new rt_scene();
    Rng g;
    g.init(seed, 0u, 0u, RT_STREAM_SCENE);
    add(s, mk3(0, -1000, 0), mk3(0, -1000, 0), 1000.0f, false, RT_MAT_LAMBERTIAN, mk3(0.5f), 0.0f);
    for (int a = -11; a < 11; a++)
        for (int b = -11; b < 11; b++) {
            float choose_mat = g.next();
            float cx = (float)a + g.next();
            float cz = (float)b + g.next();
            f3 center = mk3(cx, 0.2f, cz);
            if (choose_mat < 0.8f) {
                float r0 = g.next(), r1 = g.next(), r2 = g.next(), r3 = g.next(), r4 = g.next(), r5 = g.next();
                f3 albedo = mk3(r0 * r1, r2 * r3, r4 * r5);
                float rc = g.next();
                f3 center1 = center + mk3(0, rc * 0.5f, 0);
                add(s, center, moving ? center1 : center, 0.2f, moving, RT_MAT_LAMBERTIAN, albedo, 0.0f);
            } else if (choose_mat < 0.95f) {
                float r0 = g.next(), r1 = g.next(), r2 = g.next(), r3 = g.next();
                f3 albedo = mk3(0.5f * (1.0f + r0), 0.5f * (1.0f + r1), 0.5f * (1.0f + r2));
                add(s, center, center, 0.2f, false, RT_MAT_METAL, albedo, 0.5f * r3);
            } else {
                add(s, center, center, 0.2f, false, RT_MAT_DIELECTRIC, mk3(1.0f), 1.5f);
            }
        }
    add(s, mk3(0, 1, 0), mk3(0, 1, 0), 1.0f, false, RT_MAT_DIELECTRIC, mk3(1.0f), 1.5f);
    add(s, mk3(-4, 1, 0), mk3(-4, 1, 0), 1.0f, false, RT_MAT_LAMBERTIAN, mk3(0.4f, 0.2f, 0.1f), 0.0f);
    add(s, mk3(4, 1, 0), mk3(4, 1, 0), 1.0f, false, RT_MAT_METAL, mk3(0.7f, 0.6f, 0.5f), 0.0f);
    return s;
}
extern "C" int rt_scene_book1_final(uint64_t seed, rt_scene** out) {
    if (!out) return rt_fail(RT_ERR_INVALID, "rt_scene_book1_final: null out");
    rt_scene* s = book_scene(seed, false);
    int rc = rt_scene_build_bvh_topdown(s);  // Scenes.cu:297-299
    if (rc != RT_OK) { delete s; return rc; }
    *out = s;
    return RT_OK;
}
extern "C" int rt_scene_book2_moving(uint64_t seed, rt_scene** out) {
    if (!out) return rt_fail(RT_ERR_INVALID, "rt_scene_book2_moving: null out");
    rt_scene* s = book_scene(seed, true);
    int rc = rt_scene_build_bvh_topdown(s);
    if (rc != RT_OK) { delete s; return rc; }
    *out = s;
    return RT_OK;
}
// Book-1 three-spheres scene (BASELINE.json config 1; layout from SURVEY.md §8d) as a HittableList.
extern "C" int rt_scene_three_spheres(rt_scene** out) {
    if (!out) return rt_fail(RT_ERR_INVALID, "rt_scene_three_spheres: null out");
    rt_scene* s = new rt_scene();
    add(s, mk3(0, -100.5f, -1), mk3(0, -100.5f, -1), 100.0f, false, RT_MAT_LAMBERTIAN, mk3(0.8f, 0.8f, 0.0f), 0.0f);
    add(s, mk3(0, 0, -1.2f), mk3(0, 0, -1.2f), 0.5f, false, RT_MAT_LAMBERTIAN, mk3(0.1f, 0.2f, 0.5f), 0.0f);
    add(s, mk3(-1, 0, -1), mk3(-1, 0, -1), 0.5f, false, RT_MAT_DIELECTRIC, mk3(1.0f), 1.5f);
    add(s, mk3(-1, 0, -1), mk3(-1, 0, -1), 0.4f, false, RT_MAT_DIELECTRIC, mk3(1.0f), 1.0f / 1.5f);
    add(s, mk3(1, 0, -1), mk3(1, 0, -1), 0.5f, false, RT_MAT_METAL, mk3(0.8f, 0.6f, 0.2f), 1.0f);
    int rc = rt_scene_set_world_list(s);
    if (rc != RT_OK) { delete s; return rc; }
    *out = s;
    return RT_OK;
}

// ---- Cornell box of "Ray Tracing: The Next Week" (BASELINE.json configs[3]) ------------------------------
static void cornell_quad(rt_scene* s, f3 Q, f3 u, f3 v, int32_t mat) {
    float a[3], b[3], c[3];
    st3(a, Q); st3(b, u); st3(c, v);
    rt_scene_add_quad(s, a, b, c, mat, nullptr);
}
static f3 rot_y(f3 p, float c, float sn) { return mk3(c * p.x + sn * p.z, p.y, -sn * p.x + c * p.z); }
// box(a,b) of the book as 6 quads, rotated about y and translated on the host (the book uses rotate_y / translate instances)
static void cornell_box(rt_scene* s, f3 a, f3 b, float degrees, f3 offset, int32_t mat) {
    f3 mn = glm_min(a, b), mx = glm_max(a, b);
    f3 dx = mk3(mx.x - mn.x, 0, 0), dy = mk3(0, mx.y - mn.y, 0), dz = mk3(0, 0, mx.z - mn.z);
    float rad = radians(degrees), c = cosf(rad), sn = sinf(rad);
    f3 Qs[6] = {mk3(mn.x, mn.y, mx.z), mk3(mx.x, mn.y, mx.z), mk3(mx.x, mn.y, mn.z), mk3(mn.x, mn.y, mn.z), mk3(mn.x, mx.y, mx.z), mk3(mn.x, mn.y, mn.z)};
    f3 us[6] = {dx, -dz, -dx, dz, dx, dx};
    f3 vs[6] = {dy, dy, dy, dy, -dz, dz};
    for (int k = 0; k < 6; k++) cornell_quad(s, rot_y(Qs[k], c, sn) + offset, rot_y(us[k], c, sn), rot_y(vs[k], c, sn), mat);
}
extern "C" int rt_scene_add_box(rt_scene* s, const float a[3], const float b[3], int32_t mat, float rotate_y_degrees,
                                const float translate[3], int32_t* out_first_quad) {
    if (!s || !a || !b) return rt_fail(RT_ERR_INVALID, "rt_scene_add_box: null argument");
    if (mat < 0 || (size_t)mat >= s->mats.size()) return rt_fail(RT_ERR_INVALID, "rt_scene_add_box: material %d out of range", mat);
    if (out_first_quad) *out_first_quad = (int32_t)s->quads.size();
    cornell_box(s, ld3(a), ld3(b), rotate_y_degrees, translate ? ld3(translate) : mk3(0.0f), mat);
    return RT_OK;
}

// The image the book loads from earthmap.jpg is not redistributable here: a synthetic 256 x 128 "planet" (integer
// arithmetic only, so that the oracle generates the same bytes): oceans, land masses, polar caps.
static void synthetic_earth(std::vector<uint8_t>& img, uint32_t& w, uint32_t& h) {
    w = 256; h = 128;
    img.resize((size_t)w * h * 3);
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            uint32_t f = (x * x / 64u + y * y / 32u + x * y / 128u + 3u * x) % 64u;
            bool land = f < 26u, ice = y < 9u || y > 118u;
            uint8_t* px = &img[((size_t)y * w + x) * 3];
            if (ice) { px[0] = 235; px[1] = 240; px[2] = 245; }
            else if (land) { px[0] = (uint8_t)(60u + 2u * f); px[1] = (uint8_t)(120u + f); px[2] = 50; }
            else { px[0] = 25; px[1] = (uint8_t)(60u + f / 2u); px[2] = (uint8_t)(140u + f); }
        }
}

// final_scene() of "Ray Tracing: The Next Week" (BASELINE.json configs[4]; nothing of it exists in the reference):
// 400 ground boxes of random height (2400 quads), an area light, a moving sphere, glass, metal, a glass ball filled with a
// blue medium, a thin global fog, an image-textured sphere, a marble sphere and a rotated, translated cluster of 1000 small
// spheres, under ONE median-split BVH.  Uniforms: the host stream of rt_host_uniforms(seed), one per box, three per sphere.
extern "C" int rt_scene_book2_final(uint64_t seed, rt_scene** out) {
    if (!out) return rt_fail(RT_ERR_INVALID, "rt_scene_book2_final: null out");
    rt_scene* s = new rt_scene();
    Rng g;
    g.init(seed, 0u, 0u, RT_STREAM_SCENE);
    auto material = [&](uint32_t type, f3 albedo, float param) {
        float a[3]; st3(a, albedo);
        int32_t id = 0;
        rt_scene_add_material(s, type, a, param, nullptr, &id);
        return id;
    };
    auto sphere = [&](f3 c, float r, int32_t mat) { float p[3]; st3(p, c); add_prim(s, p, p, r, mat, false, nullptr); };
    const int32_t ground = material(RT_MAT_LAMBERTIAN, mk3(0.48f, 0.83f, 0.53f), 0.0f);
    for (int i = 0; i < 20; i++)
        for (int j = 0; j < 20; j++) {
            const float w = 100.0f;
            float x0 = -1000.0f + (float)i * w, z0 = -1000.0f + (float)j * w, y0 = 0.0f;
            float x1 = x0 + w, y1 = 1.0f + 100.0f * g.next(), z1 = z0 + w;
            cornell_box(s, mk3(x0, y0, z0), mk3(x1, y1, z1), 0.0f, mk3(0.0f), ground);
        }
    cornell_quad(s, mk3(123, 554, 147), mk3(300, 0, 0), mk3(0, 0, 265), material(RT_MAT_DIFFUSE_LIGHT, mk3(7.0f), 0.0f));
    {
        float c0[3] = {400, 400, 200}, c1[3] = {430, 400, 200};
        add_prim(s, c0, c1, 50.0f, material(RT_MAT_LAMBERTIAN, mk3(0.7f, 0.3f, 0.1f), 0.0f), true, nullptr);
    }
    const int32_t glass = material(RT_MAT_DIELECTRIC, mk3(1.0f), 1.5f);
    sphere(mk3(260, 150, 45), 50.0f, glass);
    sphere(mk3(0, 150, 145), 50.0f, material(RT_MAT_METAL, mk3(0.8f, 0.8f, 0.9f), 1.0f));
    sphere(mk3(360, 150, 145), 70.0f, glass);
    sphere(mk3(360, 150, 145), 70.0f, material(RT_MAT_ISOTROPIC, mk3(0.2f, 0.4f, 0.9f), 0.2f));
    sphere(mk3(0, 0, 0), 5000.0f, material(RT_MAT_ISOTROPIC, mk3(1.0f), 0.0001f));
    {
        uint32_t w = 0, h = 0;
        synthetic_earth(s->image, w, h);
        s->image_w = w; s->image_h = h;
    }
    sphere(mk3(400, 200, 400), 100.0f, material(RT_MAT_LAMBERTIAN_IMAGE, mk3(1.0f), 0.0f));
    rt_scene_set_perlin(s, seed);
    sphere(mk3(220, 280, 300), 80.0f, material(RT_MAT_LAMBERTIAN_NOISE, mk3(0.5f), 0.2f));
    const int32_t white = material(RT_MAT_LAMBERTIAN, mk3(0.73f), 0.0f);
    const float rad = radians(15.0f), c = cosf(rad), sn = sinf(rad);
    for (int j = 0; j < 1000; j++) {
        f3 ctr;
        ctr.x = 165.0f * g.next(); ctr.y = 165.0f * g.next(); ctr.z = 165.0f * g.next();
        sphere(rot_y(ctr, c, sn) + mk3(-100, 270, 395), 10.0f, white);   // translate(rotate_y(.., 15), (-100, 270, 395))
    }
    const float black[3] = {0.0f, 0.0f, 0.0f};
    rt_scene_set_background(s, 1, black);
    int rc = rt_scene_build_bvh_topdown(s);
    if (rc != RT_OK) { delete s; return rc; }
    *out = s;
    return RT_OK;
}

extern "C" int rt_scene_cornell_box(rt_scene** out) {
    if (!out) return rt_fail(RT_ERR_INVALID, "rt_scene_cornell_box: null out");
    rt_scene* s = new rt_scene();
    const float red[3] = {0.65f, 0.05f, 0.05f}, white[3] = {0.73f, 0.73f, 0.73f}, green[3] = {0.12f, 0.45f, 0.15f}, light[3] = {15.0f, 15.0f, 15.0f};
    int32_t m_red, m_white, m_green, m_light;
    rt_scene_add_material(s, RT_MAT_LAMBERTIAN, red, 0.0f, nullptr, &m_red);
    rt_scene_add_material(s, RT_MAT_LAMBERTIAN, white, 0.0f, nullptr, &m_white);
    rt_scene_add_material(s, RT_MAT_LAMBERTIAN, green, 0.0f, nullptr, &m_green);
    rt_scene_add_material(s, RT_MAT_DIFFUSE_LIGHT, light, 0.0f, nullptr, &m_light);
    cornell_quad(s, mk3(555, 0, 0), mk3(0, 555, 0), mk3(0, 0, 555), m_green);
    cornell_quad(s, mk3(0, 0, 0), mk3(0, 555, 0), mk3(0, 0, 555), m_red);
    cornell_quad(s, mk3(343, 554, 332), mk3(-130, 0, 0), mk3(0, 0, -105), m_light);
    cornell_quad(s, mk3(0, 0, 0), mk3(555, 0, 0), mk3(0, 0, 555), m_white);
    cornell_quad(s, mk3(555, 555, 555), mk3(-555, 0, 0), mk3(0, 0, -555), m_white);
    cornell_quad(s, mk3(0, 0, 555), mk3(555, 0, 0), mk3(0, 555, 0), m_white);
    cornell_box(s, mk3(0, 0, 0), mk3(165, 330, 165), 15.0f, mk3(265, 0, 295), m_white);
    cornell_box(s, mk3(0, 0, 0), mk3(165, 165, 165), -18.0f, mk3(130, 0, 65), m_white);
    const float black[3] = {0.0f, 0.0f, 0.0f};
    rt_scene_set_background(s, 1, black);
    int rc = rt_scene_build_bvh_topdown(s);
    if (rc != RT_OK) { delete s; return rc; }
    *out = s;
    return RT_OK;
}

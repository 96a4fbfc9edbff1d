// rt_runtime.hpp — host-side plumbing shared by the translation units of librt06.so: HIP error convention, device buffers, the device copy of a
// flat world (DeviceScene: the flat arrays + the packed LDS / global-memory image of the streaming kernels), tile map.  Internal; not installed.
#pragma once
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <algorithm>
#include <memory>
#include <mutex>
#include <string>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#include "rt06.h"
#include "rt_device_funcs.hpp"
#include "rt_internal.hpp"
#include "rt_math.hpp"
#include "rt_fastdiv.hpp"   // coord_is_regular: the fast-division class of the box coordinates
#include "rt_layout.hpp"

#define HIP_TRY(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t _e = (expr);                                                                                \
        if (_e != hipSuccess)                                                                                  \
            return rt_fail(RT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

// ---------------------------------------------------------------------------------------------
// small RAII helpers (host side)
// ---------------------------------------------------------------------------------------------
namespace {
inline float __uint_as_float_host(uint32_t u) {
    float f;
    std::memcpy(&f, &u, sizeof(f));
    return f;
}
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) {
        release();   // re-allocation (a scene packed twice) must not leak the first buffer
        const hipError_t e = hipMalloc(&p, n ? n : 1);
        if (e == hipSuccess) bytes = n; else p = nullptr;
        return e;
    }
    void release() {
        if (p) { (void)hipFree(p); p = nullptr; }
        bytes = 0;
    }
    hipError_t upload(const void* src, size_t n) {
        hipError_t e = alloc(n);
        if (e != hipSuccess) return e;
        return n ? hipMemcpy(p, src, n, hipMemcpyHostToDevice) : hipSuccess;
    }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

// one wide node of the LDS image (layout: rt_stream_kernel.hpp): per child box and axis the triple (min, max, min)
static void write_wide_node(uint4* blob, bool big, uint32_t index, const float lmin[3], const float lmax[3], const float rmin[3],
                            const float rmax[3], uint32_t lref, uint32_t rref) {
    auto bits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
    if (big) {  // one 64-byte line: [lmin.xyz lmax.x | lmax.yz rmin.xy | rmin.z rmax.xyz | left right - -]
        uint32_t* d = reinterpret_cast<uint32_t*>(blob) + (size_t)index * RT_NODE_DWORDS_BIG;
        const float v[12] = {lmin[0], lmin[1], lmin[2], lmax[0], lmax[1], lmax[2], rmin[0], rmin[1], rmin[2], rmax[0], rmax[1], rmax[2]};
        for (int k = 0; k < 12; k++) d[k] = bits(v[k]);
        d[12] = lref; d[13] = rref; d[14] = 0; d[15] = 0;
        return;
    }
    uint32_t* d = reinterpret_cast<uint32_t*>(blob) + (size_t)index * RT_NODE_DWORDS;
    for (int k = 0; k < 3; k++) {
        d[3 * k + 0] = bits(lmin[k]); d[3 * k + 1] = bits(lmax[k]); d[3 * k + 2] = bits(lmin[k]);
        d[9 + 3 * k + 0] = bits(rmin[k]); d[9 + 3 * k + 1] = bits(rmax[k]); d[9 + 3 * k + 2] = bits(rmin[k]);
    }
    d[RT_NODE_REFS] = (lref & 0xffffu) | (rref << 16);
}

struct DeviceScene {
    DevBuf nodes, prims, mats, blob, quads, perlin, image, error_flag;
    bool extended = false;  // quads, an emissive material or a constant background: beyond the reference's feature set
    bool textured = false;  // a Perlin or image material: the EXT = 2 kernels
    DeviceWorld dw{};
    PackedSceneRef packed{};  // valid when has_packed
    bool has_packed = false;
    uint32_t true_stack = 0;  // traversal-stack bound computed from the tree itself
    bool regular_boxes = false;  // all box coordinates inside the fast-division class
    bool big = false;            // packed for the global-memory kernel (the image does not fit the LDS): 64-byte nodes, breadth-first
    bool wide = false;           // ... with 32-bit references (2^14 inner nodes / 2^15 leaf codes or more); otherwise 16-bit like the LDS image
    bool any_moving = false;     // a MovingSphere is in the world: the leaf phase reads the second centres
    bool queue = false;          // RT_TRAVERSAL_QUEUE (BVH.cu:17-49's distance-sorted walk) or RT_TRAVERSAL_WIDE4: every lane walks its whole trace on its own
                                 // (the streaming kernel's RT_WORLD_BVH_QUEUE mode, records in global memory)

    // Re-pack an RT_WORLD_BVH world into the LDS image of render_kernel_stream: 76-B wide nodes (both
    // child boxes + references), 16-B sphere records, 16-B (centre1, material) records.
    // want_big: 32-bit references, records read from global memory (for worlds whose image does not fit the LDS)
    int pack(const rt_world_flat* w, bool want_big) {
        has_packed = false;
        regular_boxes = false;
        big = want_big;
        // reference width: 16 bits whenever leaf codes (< 0x7fff) and inner-node indices (< 0x4000, the bit below marks rays outside the
        // fast-division class) fit — always so for an LDS image, usually so for a BIG one
        const uint64_t n_codes64 = (uint64_t)w->n_prims * 2u + w->n_quads;
        const uint32_t n_inner_bound = w->kind == RT_WORLD_LIST ? 0u : w->n_nodes;
        const bool narrow_fits = n_codes64 < (uint64_t)RT_REF_LEAF - 1u && n_inner_bound < RT_REF_IRR;
        wide = big && (!narrow_fits || w->kind != RT_WORLD_BVH || queue || std::getenv("RT06_FORCE_WIDE") != nullptr);   // (tests force the 32-bit encoding; the queue kernels are instantiated for it)
        const uint32_t ref_leaf = wide ? RT_REF_LEAF_BIG : RT_REF_LEAF, ref_irr = wide ? RT_REF_IRR_BIG : RT_REF_IRR;
        const uint32_t sphere_codes = w->n_prims * 2u;
        if (n_codes64 >= (wide ? 0x7ffffff0ull : (uint64_t)RT_REF_LEAF - 1u) || w->n_materials > RT_MAT_INDEX_MASK) return RT_OK;  // references would not fit
        if (extended && w->kind == RT_WORLD_NODE_TREE) return RT_OK;  // quads / lights / background: BVH and HittableList worlds
        auto leaf_ref = [&](uint32_t prim) -> uint32_t {  // unified primitive index -> leaf reference
            if (prim >= w->n_prims) return ref_leaf | (sphere_codes + (prim - w->n_prims));
            return ref_leaf | (prim * 2u + ((w->prims[prim].mat & RT_PRIM_MOVING) ? 1u : 0u));
        };
        auto mat_bits = [&](uint32_t mi, uint32_t moving) -> uint32_t { return mi | (moving << 28) | (w->materials[mi].type << 29); };
        // wide nodes: BVH -> one per inner node, holding BOTH child boxes; bvh_node tree -> one per node, holding its OWN box
        std::vector<int32_t> wide_of(w->n_nodes, -1);
        uint32_t n_inner = 0;
        if (w->kind == RT_WORLD_BVH && big) {
            // breadth-first numbering: the first n_top wide nodes are the top of the tree, which the BIG kernel keeps in the LDS
            std::vector<int32_t> queue;
            if (w->nodes[w->root].left != -1) queue.push_back(w->root);
            for (size_t h = 0; h < queue.size(); h++) {
                const rt_bvh_node& n = w->nodes[queue[h]];
                wide_of[queue[h]] = (int32_t)n_inner++;
                if (w->nodes[n.left].left != -1) queue.push_back(n.left);
                if (w->nodes[n.right].left != -1) queue.push_back(n.right);
            }
        } else if (w->kind == RT_WORLD_BVH) {
            for (uint32_t i = 0; i < w->n_nodes; i++)
                if (w->nodes[i].left != -1) wide_of[i] = (int32_t)n_inner++;
        } else if (w->kind == RT_WORLD_NODE_TREE) {
            n_inner = w->n_nodes;
        }
        if (n_inner >= ref_leaf) return RT_OK;
        const uint32_t nodes_vec4 = RT_NODES_VEC4(n_inner, big);
        const uint32_t quads_at = (nodes_vec4 + w->n_prims * 2 + w->n_materials + 3u) & ~3u;   // 64-byte records on 64-byte boundaries: one cache line each
        size_t n_vec4 = (size_t)quads_at + (size_t)w->n_quads * 5;
        std::vector<uint4> host(n_vec4, make_uint4(0, 0, 0, 0));
        if (w->kind == RT_WORLD_BVH) {
            auto ref_of = [&](int32_t node) -> uint32_t {
                const rt_bvh_node& n = w->nodes[node];
                return n.left != -1 ? (uint32_t)wide_of[node] : leaf_ref((uint32_t)n.right);
            };
            for (uint32_t i = 0; i < w->n_nodes; i++) {
                if (wide_of[i] < 0) continue;
                const rt_bvh_node& n = w->nodes[i];
                const rt_bvh_node& l = w->nodes[n.left];
                const rt_bvh_node& r = w->nodes[n.right];
                write_wide_node(host.data(), big, (uint32_t)wide_of[i], l.min, l.max, r.min, r.max, ref_of(n.left), ref_of(n.right));
            }
            packed.root_ref = ref_of(w->root);
            for (int k = 0; k < 3; k++) { packed.root_min[k] = w->nodes[w->root].min[k]; packed.root_max[k] = w->nodes[w->root].max[k]; }
            // rt_fastdiv.hpp condition (a): every box coordinate is 0 or 2^-40 <= |b| < 2^40, boxes not inverted
            regular_boxes = n_inner < ref_irr;  // the fast kernel marks references with the bit below the leaf bit
            for (uint32_t i = 0; i < w->n_nodes && regular_boxes; i++)
                for (int k = 0; k < 3; k++)
                    if (!coord_is_regular(w->nodes[i].min[k]) || !coord_is_regular(w->nodes[i].max[k]) || !(w->nodes[i].min[k] <= w->nodes[i].max[k]))
                        regular_boxes = false;
        } else if (w->kind == RT_WORLD_NODE_TREE) {
            auto ref_of = [&](int32_t r) -> uint32_t { return r >= 0 ? (uint32_t)r : leaf_ref((uint32_t)(-r - 1)); };
            for (uint32_t i = 0; i < w->n_nodes; i++) {
                const rt_bvh_node& n = w->nodes[i];
                const float zero[3] = {0.0f, 0.0f, 0.0f};
                write_wide_node(host.data(), big, i, n.min, n.max, zero, zero, ref_of(n.left), ref_of(n.right));
            }
            packed.root_ref = ref_of(w->root);
            for (int k = 0; k < 3; k++) { packed.root_min[k] = w->bounds_min[k]; packed.root_max[k] = w->bounds_max[k]; }
        } else {  // HittableList: reference = RT_REF_LEAF | primitive index, pre-test against the world bounds
            packed.root_ref = ref_leaf | 0u;
            for (int k = 0; k < 3; k++) { packed.root_min[k] = w->bounds_min[k]; packed.root_max[k] = w->bounds_max[k]; }
        }
        float4* sph = reinterpret_cast<float4*>(host.data() + (size_t)nodes_vec4);
        float4* ext = sph + w->n_prims;
        any_moving = false;
        for (uint32_t i = 0; i < w->n_prims; i++) {
            const rt_prim& pr = w->prims[i];
            if (pr.mat & RT_PRIM_MOVING) any_moving = true;
            sph[i] = make_float4(pr.c0[0], pr.c0[1], pr.c0[2], pr.radius);
            uint32_t mi = pr.mat & ~RT_PRIM_MOVING;
            uint32_t moving = (pr.mat & RT_PRIM_MOVING) ? 1u : 0u;
            ext[i] = make_float4(pr.c1[0], pr.c1[1], pr.c1[2], __uint_as_float_host(mat_bits(mi, moving)));
        }
        float4* m16 = ext + w->n_prims;
        for (uint32_t i = 0; i < w->n_materials; i++) {
            const rt_material& m = w->materials[i];
            m16[i] = make_float4(m.albedo[0], m.albedo[1], m.albedo[2], m.param);
        }
        float4* qd = reinterpret_cast<float4*>(host.data()) + quads_at;
        for (uint32_t i = 0; i < w->n_quads; i++) {
            const rt_quad& q = w->quads[i];
            // 64 bytes, what quad::hit reads, in four 16-byte parts; what the shade phase reads of a quad — (normal, material) — is one
            // 16-byte record of its own behind the quads
            qd[4 * i + 0] = make_float4(q.Q[0], q.Q[1], q.Q[2], q.D);
            qd[4 * i + 1] = make_float4(q.u[0], q.u[1], q.u[2], q.v[0]);
            qd[4 * i + 2] = make_float4(q.v[1], q.v[2], q.normal[0], q.normal[1]);
            qd[4 * i + 3] = make_float4(q.normal[2], q.w[0], q.w[1], q.w[2]);
            qd[4 * (size_t)w->n_quads + i] = make_float4(q.normal[0], q.normal[1], q.normal[2], __uint_as_float_host(mat_bits(q.mat, 0u)));
        }
        HIP_TRY(blob.upload(host.data(), n_vec4 * sizeof(uint4)));
        packed.blob = blob.as<uint4>();
        packed.blob_vec4 = (uint32_t)n_vec4;
        packed.off_spheres = nodes_vec4;
        packed.off_extra = nodes_vec4 + w->n_prims;
        packed.off_mats = nodes_vec4 + w->n_prims * 2;
        packed.off_quads = quads_at;
        packed.sphere_codes = sphere_codes;
        packed.background = w->background;
        for (int k = 0; k < 3; k++) packed.background_color[k] = w->background_color[k];
        packed.n_inner = n_inner;
        packed.n_codes = sphere_codes + w->n_quads;
        packed.n_prims = w->n_prims;
        packed.n_quads = w->n_quads;
        packed.stack_cap = (true_stack ? true_stack : 1u) + 1u;  // + the sentinel entry at the bottom (RT_POP)
        packed.mats = mats.as<rt_material>();
        packed.perlin = dw.perlin; packed.image = dw.image; packed.image_w = dw.image_w; packed.image_h = dw.image_h;
        has_packed = true;
        return RT_OK;
    }
    int upload(const rt_world_flat* w) {
        if (!w) return rt_fail(RT_ERR_INVALID, "null world");
        if (w->kind > RT_WORLD_NODE_TREE) return rt_fail(RT_ERR_INVALID, "unknown world kind %u", w->kind);
        if ((w->n_prims == 0 || !w->prims) && (w->n_quads == 0 || !w->quads)) return rt_fail(RT_ERR_INVALID, "world has no primitives");
        if ((w->n_prims && !w->prims) || (w->n_quads && !w->quads)) return rt_fail(RT_ERR_INVALID, "world primitive array is null");
        if (w->background > 1) return rt_fail(RT_ERR_INVALID, "unknown background mode %u", w->background);
        if (w->traversal > RT_TRAVERSAL_WIDE4 || (w->traversal != RT_TRAVERSAL_STACK && w->kind != RT_WORLD_BVH))
            return rt_fail(RT_ERR_INVALID, "traversal mode %u: the distance-sorted queue and the 4-wide walk belong to RT_WORLD_BVH worlds", w->traversal);
        if (w->n_quads && w->kind == RT_WORLD_NODE_TREE) return rt_fail(RT_ERR_INVALID, "bvh_node trees take spheres only");
        const uint32_t n_all = w->n_prims + w->n_quads;
        if (w->n_materials == 0 || !w->materials) return rt_fail(RT_ERR_INVALID, "world has no materials");
        if (w->kind != RT_WORLD_LIST && (w->n_nodes == 0 || !w->nodes)) return rt_fail(RT_ERR_INVALID, "BVH world has no nodes");
        if (w->max_stack > RT_MAX_STACK) return rt_fail(RT_ERR_STACK, "world needs a %u-entry traversal stack; limit %d", w->max_stack, RT_MAX_STACK);
        // validate every index the kernels will follow: a bad index is a GPU fault, not an error code
        for (uint32_t i = 0; i < w->n_prims; i++)
            if ((w->prims[i].mat & ~RT_PRIM_MOVING) >= w->n_materials) return rt_fail(RT_ERR_INVALID, "primitive %u: material index out of range", i);
        for (uint32_t i = 0; i < w->n_quads; i++) {
            if (w->quads[i].mat >= w->n_materials) return rt_fail(RT_ERR_INVALID, "quad %u: material index out of range", i);
            if (w->materials[w->quads[i].mat].type == RT_MAT_ISOTROPIC) return rt_fail(RT_ERR_INVALID, "quad %u: a constant medium is bounded by a sphere (RT_MAT_ISOTROPIC on a quad)", i);
        }
        extended = w->n_quads != 0 || w->background != 0;
        textured = false;
        for (uint32_t i = 0; i < w->n_materials; i++) {
            if (w->materials[i].type > RT_MAT_LAMBERTIAN_IMAGE) return rt_fail(RT_ERR_INVALID, "material %u: unknown type", i);
            if (w->materials[i].type == RT_MAT_LAMBERTIAN_NOISE && !w->perlin) return rt_fail(RT_ERR_INVALID, "material %u is a noise texture but the world has no Perlin tables (rt_scene_set_perlin)", i);
            if (w->materials[i].type == RT_MAT_LAMBERTIAN_IMAGE && (!w->image || w->image_width == 0 || w->image_height == 0))
                return rt_fail(RT_ERR_INVALID, "material %u is an image texture but the world has no image (rt_scene_set_image)", i);
            if (w->materials[i].type == RT_MAT_ISOTROPIC && !(w->materials[i].param > 0.0f)) return rt_fail(RT_ERR_INVALID, "material %u: a constant medium needs a density > 0", i);
            if (w->materials[i].type >= RT_MAT_DIFFUSE_LIGHT) extended = true;
            if (w->materials[i].type >= RT_MAT_LAMBERTIAN_NOISE) textured = true;
        }
        if (w->kind == RT_WORLD_BVH) {
            if (w->root < 0 || (uint32_t)w->root >= w->n_nodes) return rt_fail(RT_ERR_INVALID, "BVH root out of range");
            for (uint32_t i = 0; i < w->n_nodes; i++) {
                const rt_bvh_node& n = w->nodes[i];
                if (n.left == -1) {
                    if (n.right < 0 || (uint32_t)n.right >= n_all) return rt_fail(RT_ERR_INVALID, "BVH leaf %u: primitive index out of range", i);
                } else if (n.left < 0 || (uint32_t)n.left >= w->n_nodes || n.right < 0 || (uint32_t)n.right >= w->n_nodes || (uint32_t)n.left == i || (uint32_t)n.right == i)
                    return rt_fail(RT_ERR_INVALID, "BVH node %u: child index out of range", i);
            }
        } else if (w->kind == RT_WORLD_NODE_TREE) {
            auto ok = [&](int32_t r) { return r >= 0 ? (uint32_t)r < w->n_nodes : (uint32_t)(-r - 1) < w->n_prims; };
            if (!ok(w->root)) return rt_fail(RT_ERR_INVALID, "bvh_node tree root out of range");
            for (uint32_t i = 0; i < w->n_nodes; i++)
                if (!ok(w->nodes[i].left) || !ok(w->nodes[i].right) || w->nodes[i].left == (int32_t)i || w->nodes[i].right == (int32_t)i)
                    return rt_fail(RT_ERR_INVALID, "bvh_node %u: child reference out of range", i);
        }
        // the node graph must be a tree no deeper than the traversal stack: a cycle would spin the GPU
        // forever and a deeper tree would overrun the per-lane stack (the reference checks neither).
        true_stack = 0;
        if (w->kind != RT_WORLD_LIST) {
            std::vector<uint8_t> seen(w->n_nodes, 0);
            std::vector<std::pair<int32_t, uint32_t>> todo;  // (node, depth)
            auto is_node = [&](int32_t r) { return w->kind == RT_WORLD_BVH ? true : r >= 0; };
            uint32_t max_leaf_depth = 0;
            if (is_node(w->root)) todo.push_back({w->root, 0u});
            while (!todo.empty()) {
                auto [ni, d] = todo.back();
                todo.pop_back();
                if (seen[ni]) return rt_fail(RT_ERR_INVALID, "node %d is reachable twice: the node graph is not a tree", ni);
                seen[ni] = 1;
                const rt_bvh_node& n = w->nodes[ni];
                if (w->kind == RT_WORLD_BVH && n.left == -1) { max_leaf_depth = std::max(max_leaf_depth, d); continue; }
                max_leaf_depth = std::max(max_leaf_depth, d + 1);
                if (is_node(n.left)) todo.push_back({n.left, d + 1});
                if (is_node(n.right)) todo.push_back({n.right, d + 1});
            }
            true_stack = max_leaf_depth + 1;
            if (true_stack > RT_MAX_STACK)
                return rt_fail(RT_ERR_STACK, "world needs a %u-entry traversal stack; limit %d (BVH.cu:17)", true_stack, RT_MAX_STACK);
        }
        HIP_TRY(nodes.upload(w->nodes, sizeof(rt_bvh_node) * (size_t)w->n_nodes));
        HIP_TRY(prims.upload(w->prims, sizeof(rt_prim) * (size_t)w->n_prims));
        HIP_TRY(mats.upload(w->materials, sizeof(rt_material) * (size_t)w->n_materials));
        HIP_TRY(quads.upload(w->quads, sizeof(rt_quad) * (size_t)w->n_quads));
        dw.kind = w->kind; dw.root = w->root;
        dw.n_nodes = w->n_nodes; dw.n_prims = w->n_prims; dw.n_mats = w->n_materials;
        dw.bmin = mk3(w->bounds_min[0], w->bounds_min[1], w->bounds_min[2]);
        dw.bmax = mk3(w->bounds_max[0], w->bounds_max[1], w->bounds_max[2]);
        dw.nodes = nodes.as<rt_bvh_node>(); dw.prims = prims.as<rt_prim>(); dw.mats = mats.as<rt_material>();
        dw.quads = quads.as<rt_quad>(); dw.n_quads = w->n_quads;
        dw.background = w->background;
        dw.background_color = mk3(w->background_color[0], w->background_color[1], w->background_color[2]);
        HIP_TRY(perlin.upload(w->perlin, w->perlin ? sizeof(rt_perlin) : 0));
        HIP_TRY(image.upload(w->image, w->image ? (size_t)w->image_width * w->image_height * 3 : 0));
        dw.perlin = w->perlin ? perlin.as<rt_perlin>() : nullptr;
        dw.image = w->image ? image.as<uint8_t>() : nullptr;
        dw.image_w = w->image_width; dw.image_h = w->image_height;
        dw.traversal = w->traversal;
        HIP_TRY(error_flag.alloc(4));
        HIP_TRY(hipMemset(error_flag.p, 0, 4));
        dw.error_flag = error_flag.as<uint32_t>();
        // 16-bit references and an LDS-resident image when that fits (2 x 768-thread workgroups per CU want <= 80 KiB each,
        // one workgroup may take all 160 KiB); otherwise 32-bit references and the records stay in global memory / L2
        queue = w->traversal != RT_TRAVERSAL_STACK;   // the queue or the 4-wide walk: every lane walks its trace on its own
        if (queue) return pack(w, true);   // the queue walk reads the flat world itself; the shade phase reads the packed records from global memory
        int rc = pack(w, false);
        if (rc != RT_OK) return rc;
        const bool fits_lds = has_packed && (size_t)packed.blob_vec4 * 16u + (size_t)RT_STREAM_BLOCK * packed.stack_cap * 2u <= 160u * 1024u;
        const char* force = std::getenv("RT06_FORCE_BIG");  // measurements / tests: take the global-memory path for any BVH world
        if (!fits_lds || (force && force[0] == '1')) rc = pack(w, true);
        return rc;
    }
};

// render_kernel_xchg: did a bounded ring wait run out?  Called after a synchronisation.
int check_xchg_error(DevBuf& flag_buf) {
    if (!flag_buf.p) return RT_OK;
    uint32_t flag = 0;
    HIP_TRY(hipMemcpy(&flag, flag_buf.p, 4, hipMemcpyDeviceToHost));
    if (flag) {
        if (std::getenv("RT06_DEBUG")) {   // what the waves that gave up saw
            std::vector<uint32_t> h(flag_buf.bytes / 4u);
            HIP_TRY(hipMemcpy(h.data(), flag_buf.p, flag_buf.bytes, hipMemcpyDeviceToHost));
            int shown = 0;
            for (size_t w = 0; 16u + (w + 1) * RT_XCHG_DEBUG_WORDS <= h.size() && shown < 60; w++) {
                const uint32_t* o = h.data() + 16u + w * RT_XCHG_DEBUG_WORDS;
                if (!o[0]) continue;
                shown++;
                fprintf(stderr, "[rt06 xchg] wg %zu wave %zu why %u: a %u b %u c %u d %u | sq h %u t %u tq h %u t %u | pop %u dry %u done %u err %u\n", w / (RT_XCHG_BLOCK / 64u),
                        w % (RT_XCHG_BLOCK / 64u), o[0], o[1], o[2], o[3], o[4], o[5] & 0xffffu, o[5] >> 16, o[6] & 0xffffu, o[6] >> 16, o[7], o[8], o[9], o[10]);
            }
        }
        HIP_TRY(hipMemset(flag_buf.p, 0, flag_buf.bytes));
        return rt_fail(RT_ERR_HIP, "render_kernel_xchg: a wait ran out of its bound (code %u, exchange protocol failure): the frame is incomplete", flag);
    }
    return RT_OK;
}

// RT_TRAVERSAL_QUEUE: has a lane overflowed the 32-entry queue?  Called after a synchronisation.
int check_traversal_overflow(DeviceScene& sc) {
    if (sc.dw.traversal == RT_TRAVERSAL_STACK) return RT_OK;
    uint32_t flag = 0;
    HIP_TRY(hipMemcpy(&flag, sc.error_flag.p, 4, hipMemcpyDeviceToHost));
    if (flag) {
        HIP_TRY(hipMemset(sc.error_flag.p, 0, 4));
        return rt_fail(RT_ERR_STACK, "the %s overflowed its %d entries (BVH.cu:17): the results are incomplete",
                       sc.dw.traversal == RT_TRAVERSAL_QUEUE ? "distance-sorted traversal queue" : "stack of the 4-wide walk", RT_MAX_STACK);
    }
    return RT_OK;
}

int select_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
        return rt_fail(RT_ERR_NO_DEVICE, "no HIP device available: the HIP path is required, there is no CPU fallback");
    if (device < 0 || device >= n) return rt_fail(RT_ERR_INVALID, "device %d out of range (%d devices)", device, n);
    HIP_TRY(hipSetDevice(device));
    return RT_OK;
}
}  // namespace

// Pixel ownership: 8x8 tiles in row-major tile order, tile t belongs to rank t % world_size; a rank's shard
// is tile-major and has the same size on every rank (the last tiles may be padding).
static TileMap make_tile_map(uint32_t width, uint32_t height, uint32_t rank, uint32_t world_size) {
    TileMap tm{};
    tm.width = width; tm.height = height;
    tm.tiles_x = (width + RT_TILE - 1) / RT_TILE;
    const uint32_t tiles_y = (height + RT_TILE - 1) / RT_TILE;
    tm.n_tiles = tm.tiles_x * tiles_y;
    tm.rank = rank; tm.world_size = world_size;
    tm.n_local_tiles = (tm.n_tiles + world_size - 1) / world_size;
    tm.direct = world_size == 1 ? 1u : 0u;
    return tm;
}

// what the multi-GPU driver (rt_multi.hip) needs of a renderer beyond the C ABI; defined in rt_device.hip
hipStream_t rt_renderer_own_stream(rt_renderer* r);      // the renderer's non-blocking stream
float* rt_renderer_own_framebuffer(rt_renderer* r);      // its device framebuffer (world_size > 1: the tile-major shard)
int rt_renderer_check_device_flags(rt_renderer* r);      // after a synchronisation: ray-exchange protocol error / traversal-queue overflow

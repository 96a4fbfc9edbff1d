// rt_device.hip — HIP kernels (gfx950) and the device half of the C ABI: Renderer + probes.
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
#include "rt_render_kernels.hpp"
#include "rt_stream_kernel.hpp"
#include "rt_xchg_kernel.hpp"

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
#ifdef RT_BIG_TRIPLES
    if (big) {
        uint32_t* d = reinterpret_cast<uint32_t*>(blob) + (size_t)index * RT_NODE_DWORDS_BIG;
        for (int k = 0; k < 3; k++) {
            d[3 * k + 0] = bits(lmin[k]); d[3 * k + 1] = bits(lmax[k]); d[3 * k + 2] = bits(lmin[k]);
            d[9 + 3 * k + 0] = bits(rmin[k]); d[9 + 3 * k + 1] = bits(rmax[k]); d[9 + 3 * k + 2] = bits(rmin[k]);
        }
        d[18] = lref; d[19] = rref;
        return;
    }
#endif
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
#ifdef RT_BRANCHLESS_STACK
        packed.stack_cap += 1u;  // the unconditional far-child store may touch one entry above the deepest push
#endif
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

extern "C" int rt_device_info(int device, uint32_t out[4]) {
    if (!out) return rt_fail(RT_ERR_INVALID, "rt_device_info: null out");
    int rc = select_device(device);
    if (rc != RT_OK) return rc;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    out[0] = (uint32_t)prop.multiProcessorCount;      // compute units
    out[1] = (uint32_t)prop.clockRate;                 // peak engine clock, kHz
    out[2] = (uint32_t)(prop.totalGlobalMem >> 20);   // HBM, MiB
    out[3] = (uint32_t)prop.memoryClockRate;           // kHz
    return RT_OK;
}

extern "C" int rt_shard_layout(uint32_t width, uint32_t height, uint32_t world_size, uint32_t out[4]) {
    if (!out || width == 0 || height == 0 || world_size == 0) return rt_fail(RT_ERR_INVALID, "rt_shard_layout: bad argument");
    const TileMap tm = make_tile_map(width, height, 0, world_size);
    out[0] = tm.tiles_x; out[1] = tm.n_tiles; out[2] = tm.n_local_tiles; out[3] = tm.n_local_tiles * RT_TILE * RT_TILE * 4u;
    return RT_OK;
}

extern "C" int rt_shard_pixel_map(uint32_t width, uint32_t height, uint32_t world_size, uint32_t rank, uint32_t* out_gid, size_t n) {
    if (!out_gid || width == 0 || height == 0 || world_size == 0 || rank >= world_size) return rt_fail(RT_ERR_INVALID, "rt_shard_pixel_map: bad argument");
    const TileMap tm = make_tile_map(width, height, rank, world_size);
    if (n != (size_t)tm.n_local_tiles * RT_TILE * RT_TILE) return rt_fail(RT_ERR_INVALID, "rt_shard_pixel_map: a shard has %u pixels", tm.n_local_tiles * RT_TILE * RT_TILE);
    for (uint32_t L = 0; L < (uint32_t)n; L++) {   // the host statement of local_pixel_to_gid (csrc/rt_render_kernels.hpp)
        const uint32_t tl = L / (RT_TILE * RT_TILE), p = L % (RT_TILE * RT_TILE);
        const uint32_t gt = tl * world_size + rank;
        const uint32_t x = (gt % tm.tiles_x) * RT_TILE + (p % RT_TILE), y = (gt / tm.tiles_x) * RT_TILE + (p / RT_TILE);
        out_gid[L] = (gt < tm.n_tiles && x < width && y < height) ? y * width + x : 0xffffffffu;
    }
    return RT_OK;
}

extern "C" int rt_device_count(int* out) {
    if (!out) return rt_fail(RT_ERR_INVALID, "rt_device_count: null out");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *out = n;
    return RT_OK;
}

// ---------------------------------------------------------------------------------------------
// Renderer
// ---------------------------------------------------------------------------------------------
struct rt_renderer {
    rt_render_config cfg{};
    rt_camera cam{};
    DeviceScene scene;
    TileMap tm{};
    DevBuf fb;
    DevBuf work_counter;
    DevBuf samples, running;     // sample buffer of one pass (12 B per sample); running sums (16 B per pixel) when spp needs several passes
    // primary rays of one pass: 3 arrays of 16 B per sample index (origin|time, direction, RNG state).  Generated on the render's
    // own stream, before the streaming kernel: generating pass k + 1 on a second stream WHILE pass k is traced was measured and is
    // harmful (the persistent kernel ran 40 % slower with the generator's waves co-resident: 100 ms instead of 70).
    DevBuf primary[1];
    uint32_t pass_spp = 0;       // samples per pixel per pass
    uint32_t n_cus = 0;
    uint32_t stream_lds_bytes = 0;
    uint32_t n_top = 0;  // BIG kernels: wide nodes (breadth-first order) staged in the LDS
    uint32_t stream_block = RT_STREAM_BLOCK;
    uint32_t stream_blocks_per_cu = 0;
    uint32_t variant = 0;        // resolved kernel variant (see rt_render_config::variant)
    uint32_t tol = 0;            // 1 / 2: variant 3 with the tolerance-mode box test (requested as variant 6 / 7)
    uint32_t tune[3] = {RT_INNER_KEEP, RT_SHADE_MIN, RT_LEAF_MIN};  // scheduling thresholds of the streaming kernel
    // render_kernel_xchg (variant 5): roles, ring capacities, population and thresholds (RT06_XCHG=tracers,extra,swap,shade,patience,prio)
    struct { uint32_t n_tracers = 9, tq_cap = 0, sq_cap = 0, pop_extra = 192, swap_min = 16, shade_min = 48, patience = 6, prio = 1, scene_vec4 = 0, extra_in_lds = 0, keep = 44, shards = 1; } xc;
    DevBuf xchg_error;           // set by the kernel when a bounded ring wait ran out (a protocol bug, never expected)
    size_t shard_floats = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    // per-kernel HIP events of the last RT_TIMES_RING render calls, on the stream the kernels run on, four per PASS:
    // [4k] before primary_rays_kernel of pass k, [4k+1] before the streaming kernel, [4k+2] after it, [4k+3] after resolve_kernel.
    // Created at first use (a 10 000-spp render of a 4K frame has > 100 passes).
    static constexpr uint32_t RT_TIMES_RING = 32;
    std::vector<hipEvent_t> kev[RT_TIMES_RING];
    uint64_t n_renders = 0;
    uint32_t n_passes = 1;
    static constexpr uint32_t SAMPLE_BYTES = RT_SAMPLE_BYTES, PRIMARY_BYTES = 48;   // HBM per sample index of a pass: radiance (float4) + primary ray record

    // Pick the kernel variant and size the per-pass sample buffer.
    //   0 = default (the fastest validated variant), 1 = baseline wave-per-pixel kernel,
    //   2 = streaming kernel with verbatim box tests, 3 = streaming kernel with the fast exact division,
    //   4 = 3 + filtered box-pair predicates (experimental).
    int plan() {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, cfg.device));
        n_cus = (uint32_t)prop.multiProcessorCount;
        const uint32_t lds_per_cu = 160u * 1024u;  // MI355X_MICROARCH.md: 160 KiB LDS per CU
        uint32_t want = cfg.variant;
        if (want > 7) return rt_fail(RT_ERR_INVALID, "unknown kernel variant %u", want);
        const uint32_t tol_form = want >= 6 ? want - 5u : 0u;   // variants 6 / 7: variant 3 with the tolerance-mode box test (NOT bit-exact; opt-in measurement)
        if (tol_form) want = 3;
        bool can_stream = scene.has_packed;
        if (can_stream) {
            stream_block = RT_STREAM_BLOCK;
            if (const char* env = std::getenv("RT06_BLOCK")) {  // occupancy experiments: 512 / 768 / 1024 threads — instantiated for the
                int v = std::atoi(env);                         // reference-feature BVH kernel (variant 3) only; every other kernel is 768
                const bool has_instances = !scene.big && !scene.extended && scene.dw.kind == RT_WORLD_BVH && scene.regular_boxes && (cfg.variant == 0 || cfg.variant == 3);
                if ((v == 512 || v == 768 || v == 1024) && has_instances) stream_block = (uint32_t)v;
            }
            if (scene.big) {  // the per-lane stacks (32-bit entries) and, in what two workgroups per CU leave free, the top of the tree
                stream_block = RT_STREAM_BLOCK;
                const uint32_t stacks = (stream_block * scene.packed.stack_cap * (scene.wide ? 4u : 2u) + 63u) & ~63u;
                const uint32_t budget = stacks + 4096u <= lds_per_cu / 2u ? lds_per_cu / 2u : lds_per_cu;
                uint32_t top_bytes = budget > stacks ? budget - stacks : 0u;
                top_bytes = std::min(top_bytes & ~63u, scene.packed.n_inner * (RT_NODE_DWORDS_BIG * 4u));
                if (const char* env = std::getenv("RT06_TOP_NODES")) top_bytes = std::min(top_bytes, (uint32_t)std::atoi(env) * (RT_NODE_DWORDS_BIG * 4u));
                if (scene.queue) top_bytes = 0;   // the queue walk reads the flat world's own nodes
#ifdef RT_BIG_TRIPLES
                top_bytes = 0;
#endif
                n_top = top_bytes / (RT_NODE_DWORDS_BIG * 4u);
                stream_lds_bytes = top_bytes + stacks;
            } else {
                // (1024-thread workgroups keep the material records in global memory: rt_stream_kernel.hpp MATS_GLOBAL)
                stream_lds_bytes = (stream_block == 1024u ? scene.packed.off_mats : scene.packed.blob_vec4) * 16u + stream_block * scene.packed.stack_cap * 2u;
                stream_lds_bytes = (stream_lds_bytes + 15u) & ~15u;
            }
            if (stream_lds_bytes > lds_per_cu) can_stream = false;
            else stream_blocks_per_cu = std::min(2u, lds_per_cu / stream_lds_bytes);
        }
        const bool can_xchg = can_stream && !scene.big && !scene.extended && scene.dw.kind == RT_WORLD_BVH && scene.regular_boxes && !scene.queue;
        if (scene.queue && want >= 3)
            return rt_fail(RT_ERR_INVALID, "kernel variants 3 to 5 walk the tree with the stack of BVH.cu:54-106: a world with another traversal rule (RT_TRAVERSAL_QUEUE, RT_TRAVERSAL_WIDE4) renders on variant 2 (or 0) and on the baseline kernel (1)");
        if (want == 0) want = can_stream ? ((scene.dw.kind == RT_WORLD_BVH && scene.regular_boxes && !scene.queue) ? 3u : 2u) : 1u;
        if (want == 3 && cfg.variant == 0 && can_xchg) {
            const char* env = std::getenv("RT06_DEFAULT_XCHG");
            if (env && env[0] == '1') want = 5;
        }
        if (want == 5 && !can_xchg)
            return rt_fail(RT_ERR_INVALID, "kernel variant 5 (ray exchange) needs an LDS-resident RT_WORLD_BVH world of the reference's feature set with box coordinates in the fast-division class");
        if (want == 4 && can_stream && scene.big)
            return rt_fail(RT_ERR_INVALID, "kernel variant 4 needs a world whose LDS image fits in 160 KiB: use variant 0, 2 or 3");
        if (want == 4 && scene.extended)
            return rt_fail(RT_ERR_INVALID, "kernel variant 4 renders the reference's feature set only (no quads / lights / constant background): use variant 0, 2 or 3");
        if (want >= 3 && want <= 4 && scene.dw.kind != RT_WORLD_BVH)
            return rt_fail(RT_ERR_INVALID, "kernel variants 3 and 4 need an RT_WORLD_BVH world (a HittableList / bvh_node world runs on variant 2)");
        if (want >= 2 && !can_stream)
            return rt_fail(RT_ERR_INVALID, "kernel variant %u cannot take this world (its references or per-lane stacks do not fit)", want);
        if (want >= 3 && !scene.regular_boxes)
            return rt_fail(RT_ERR_INVALID, "kernel variants 3 and 4 need every box coordinate to be 0 or within [2^-40, 2^40)");
        variant = want;
        if (tol_form) {
            if (scene.big || scene.extended || stream_block != RT_STREAM_BLOCK)
                return rt_fail(RT_ERR_INVALID, "kernel variants 6 and 7 (tolerance-mode box test) are instantiated for LDS-resident worlds of the reference's feature set only");
            tol = tol_form;
        }
        if (variant == 5) {
            // LDS of a workgroup (two per CU): nodes | spheres | (second centres when a sphere moves) | tracer stacks | rings
            stream_block = RT_XCHG_BLOCK;
            if (const char* env = std::getenv("RT06_XCHG")) {
                unsigned v[8] = {xc.n_tracers, xc.pop_extra, xc.swap_min, xc.shade_min, xc.patience, xc.prio, xc.keep, xc.shards};
                const int n = std::sscanf(env, "%u,%u,%u,%u,%u,%u,%u,%u", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6], &v[7]);
                if (n >= 1 && v[0] >= 1 && v[0] <= RT_XCHG_BLOCK / 64 - 1) xc.n_tracers = v[0];
                if (n >= 2 && v[1] <= 1024) xc.pop_extra = v[1];
                if (n >= 3 && v[2] >= 1 && v[2] <= 64) xc.swap_min = v[2];
                if (n >= 4 && v[3] >= 1 && v[3] <= 64) xc.shade_min = v[3];
                if (n >= 5 && v[4] <= 1000) xc.patience = v[4];
                if (n >= 6) xc.prio = v[5] ? 1u : 0u;
                if (n >= 7 && v[6] >= 1 && v[6] <= 64) xc.keep = v[6];
                if (n >= 8 && (v[7] == 1 || v[7] == 2)) xc.shards = v[7];
            }
            if (xc.shards > xc.n_tracers || xc.shards > RT_XCHG_BLOCK / 64 - xc.n_tracers) xc.shards = 1;   // every shard needs a tracer and a shader
            xc.extra_in_lds = scene.any_moving ? 1u : 0u;
            xc.scene_vec4 = scene.any_moving ? scene.packed.off_mats : scene.packed.off_extra;
            const uint32_t fixed = xc.scene_vec4 * 16u + ((xc.n_tracers * 64u * scene.packed.stack_cap * 2u + 15u) & ~15u) + xc.shards * XC_WORDS * 4u;
            static const uint32_t caps[][2] = {{128, 128}, {64, 128}, {64, 64}, {32, 64}, {32, 32}, {16, 32}, {16, 16}};   // per workgroup: divided by the shards
            xc.tq_cap = 0;
            for (const auto& c : caps) {
                const uint32_t total = fixed + c[0] * (4u + XC_TQ_ENTRY_BYTES) + c[1] * (4u + XC_SQ_ENTRY_BYTES);
                if (total <= lds_per_cu / 2u && c[0] / xc.shards >= 16u) { xc.tq_cap = c[0] / xc.shards; xc.sq_cap = c[1] / xc.shards; stream_lds_bytes = (total + 15u) & ~15u; break; }
            }
            if (xc.tq_cap == 0) {
                if (cfg.variant == 5) return rt_fail(RT_ERR_INVALID, "kernel variant 5: the scene image leaves no room for the ray rings in the LDS");
                variant = 3;   // chosen by default only: fall back to the streaming kernel
            } else {
                stream_blocks_per_cu = 2;
                // the population must stay below what the places that can hold a ray add up to (no full-ring deadlock)
                xc.pop_extra = std::min(xc.pop_extra, xc.shards * (xc.tq_cap + xc.sq_cap - 16u));
                const size_t err_bytes = 64u + (size_t)n_cus * 2u * (RT_XCHG_BLOCK / 64u) * RT_XCHG_DEBUG_WORDS * 4u;
                HIP_TRY(xchg_error.alloc(err_bytes));
                HIP_TRY(hipMemset(xchg_error.p, 0, err_bytes));
            }
        }
        if ((variant == 2 || variant == 4) && !scene.big) {
            stream_block = 768;
            stream_lds_bytes = (scene.packed.blob_vec4 * 16u + stream_block * scene.packed.stack_cap * 2u + 15u) & ~15u;
            stream_blocks_per_cu = std::min(2u, lds_per_cu / stream_lds_bytes);
        }
        if (const char* env = std::getenv("RT06_TUNE")) {  // "keep,shade,leaf" — scheduling experiments only; results never change
            unsigned a = 0, b = 0, c = 0;
            if (std::sscanf(env, "%u,%u,%u", &a, &b, &c) == 3 && a >= 1 && a <= 64 && b >= 1 && b <= 64 && c >= 1 && c <= 64) {
                tune[0] = a; tune[1] = b; tune[2] = c;
            }
        }
        if (variant >= 2) {
            // HBM of one pass: every sample index owns SAMPLE_BYTES of radiance + PRIMARY_BYTES of primary-ray record.  The default
            // budget — 120 GiB of the 288, but never more than 45 % of what is free on the device right now, so that two renderers of a
            // big frame can live side by side — gives the 1200x800x500 headline one pass (28.8 GB) and a 3840x2160 frame 258 spp per
            // pass (40 GiB, round 2's default, gave 86: 117 passes instead of 39 for 10 000 spp cost 1.1 % in per-pass tails).
            uint64_t budget = 120ull << 30;
            {
                size_t free_b = 0, total_b = 0;
                if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > 0) budget = std::min<uint64_t>(budget, (uint64_t)free_b / 100u * 45u);
            }
            const uint64_t per_sample = SAMPLE_BYTES + PRIMARY_BYTES;
            if (const char* env = std::getenv("RT06_PASS_BUDGET_BYTES")) {  // bytes of ALL per-sample buffers of a pass
                unsigned long long v = std::strtoull(env, nullptr, 10);
                if (v >= per_sample) budget = v;
            }
            uint64_t n_local_pixels = (uint64_t)tm.n_local_tiles * RT_TILE * RT_TILE;
            uint64_t max_spp = std::max<uint64_t>(1, budget / (n_local_pixels * per_sample));
            if (const char* env = std::getenv("RT06_PASS_SPP")) {  // tests force multi-pass rendering with this
                unsigned long long v = std::strtoull(env, nullptr, 10);
                if (v >= 1) max_spp = v;
            }
            max_spp = std::min<uint64_t>(max_spp, (0xF0000000ull - 1) / n_local_pixels);   // sample indices of a pass are 32 bits wide
            if (max_spp == 0) return rt_fail(RT_ERR_INVALID, "image too large for one pass");
            pass_spp = (uint32_t)std::min<uint64_t>(cfg.samples_per_pixel, max_spp);
            for (;;) {   // a device that cannot give the pass its buffers gets smaller passes, not an error: halve until they fit
                hipError_t e = samples.alloc((size_t)(n_local_pixels * pass_spp * SAMPLE_BYTES));
                if (e == hipSuccess) e = primary[0].alloc((size_t)(n_local_pixels * pass_spp * PRIMARY_BYTES));
                if (e == hipSuccess) break;
                (void)hipGetLastError();   // (clears the sticky out-of-memory status)
                samples.release(); primary[0].release();
                if (e != hipErrorOutOfMemory || pass_spp == 1u)
                    return rt_fail(RT_ERR_HIP, "per-pass buffers (%llu bytes per sample index x %llu sample indices): %s", (unsigned long long)per_sample,
                                   (unsigned long long)(n_local_pixels * pass_spp), hipGetErrorString(e));
                pass_spp = (pass_spp + 1u) / 2u;
            }
            n_passes = (cfg.samples_per_pixel + pass_spp - 1) / pass_spp;
            if (n_passes > 1) HIP_TRY(running.alloc((size_t)(n_local_pixels * 16ull)));
            HIP_TRY(hipFuncSetAttribute(stream_kernel_ptr(), hipFuncAttributeMaxDynamicSharedMemorySize, (int)stream_lds_bytes));
            if (std::getenv("RT06_DEBUG")) {
                int occ = -1;
                (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, stream_kernel_ptr(), (int)stream_block, stream_lds_bytes);
                fprintf(stderr, "[rt06] stream kernel: block %u, LDS %u B, planned %u blocks/CU, runtime occupancy query %d blocks/CU\n",
                        stream_block, stream_lds_bytes, stream_blocks_per_cu, occ);
            }
        }
        return RT_OK;
    }

    const void* stream_kernel_ptr() const {
        if (variant == 5) return reinterpret_cast<const void*>(&render_kernel_xchg<RT_XCHG_BLOCK>);
        const bool fast = variant == 3;
        if (scene.queue) {   // the distance-sorted queue / the 4-wide walk: one instantiation per feature level, records in global memory, 32-bit references
            if (scene.textured) return reinterpret_cast<const void*>(&render_kernel_stream<true, false, 768, RT_WORLD_BVH_QUEUE, 2, true, true>);
            if (scene.extended) return reinterpret_cast<const void*>(&render_kernel_stream<true, false, 768, RT_WORLD_BVH_QUEUE, 1, true, true>);
            return reinterpret_cast<const void*>(&render_kernel_stream<true, false, 768, RT_WORLD_BVH_QUEUE, 0, true, true>);
        }
        if (scene.big) {   // records in global memory: <EXACT, FILTER, BLOCK, WORLD, EXT, BIG = true, WIDE>
#define RT_BIG_KERNEL(exact, world, ext, wide_) reinterpret_cast<const void*>(&render_kernel_stream<exact, false, 768, world, ext, true, wide_>)
            if (scene.dw.kind == RT_WORLD_LIST) return scene.textured ? RT_BIG_KERNEL(true, RT_WORLD_LIST, 2, true) : RT_BIG_KERNEL(true, RT_WORLD_LIST, 1, true);
            if (scene.dw.kind == RT_WORLD_NODE_TREE) return RT_BIG_KERNEL(true, RT_WORLD_NODE_TREE, 0, true);
            if (scene.wide) {
                if (scene.textured) return fast ? RT_BIG_KERNEL(false, RT_WORLD_BVH, 2, true) : RT_BIG_KERNEL(true, RT_WORLD_BVH, 2, true);
                return fast ? RT_BIG_KERNEL(false, RT_WORLD_BVH, 1, true) : RT_BIG_KERNEL(true, RT_WORLD_BVH, 1, true);
            }
            if (scene.textured) return fast ? RT_BIG_KERNEL(false, RT_WORLD_BVH, 2, false) : RT_BIG_KERNEL(true, RT_WORLD_BVH, 2, false);
            return fast ? RT_BIG_KERNEL(false, RT_WORLD_BVH, 1, false) : RT_BIG_KERNEL(true, RT_WORLD_BVH, 1, false);
#undef RT_BIG_KERNEL
        }
        if (scene.dw.kind == RT_WORLD_LIST && scene.extended)
            return scene.textured ? reinterpret_cast<const void*>(&render_kernel_stream<true, false, 768, RT_WORLD_LIST, 2>)
                                  : reinterpret_cast<const void*>(&render_kernel_stream<true, false, 768, RT_WORLD_LIST, 1>);
        if (scene.extended) {
            if (scene.textured) return fast ? reinterpret_cast<const void*>(&render_kernel_stream<false, false, 768, RT_WORLD_BVH, 2>)
                                            : reinterpret_cast<const void*>(&render_kernel_stream<true, false, 768, RT_WORLD_BVH, 2>);
            return fast ? reinterpret_cast<const void*>(&render_kernel_stream<false, false, 768, RT_WORLD_BVH, 1>)
                        : reinterpret_cast<const void*>(&render_kernel_stream<true, false, 768, RT_WORLD_BVH, 1>);
        }
        if (scene.dw.kind == RT_WORLD_LIST) return reinterpret_cast<const void*>(&render_kernel_stream<true, false, 768, RT_WORLD_LIST>);
        if (scene.dw.kind == RT_WORLD_NODE_TREE) return reinterpret_cast<const void*>(&render_kernel_stream<true, false, 768, RT_WORLD_NODE_TREE>);
        if (tol == 1) return reinterpret_cast<const void*>(&render_kernel_stream<false, false, 768, RT_WORLD_BVH, 0, false, false, 1>);
        if (tol == 2) return reinterpret_cast<const void*>(&render_kernel_stream<false, false, 768, RT_WORLD_BVH, 0, false, false, 2>);
        if (variant == 2) return reinterpret_cast<const void*>(&render_kernel_stream<true, false, 768>);
        if (variant == 4) return reinterpret_cast<const void*>(&render_kernel_stream<false, true, 768>);
        if (stream_block == 512) return reinterpret_cast<const void*>(&render_kernel_stream<false, false, 512>);
        if (stream_block == 1024) return reinterpret_cast<const void*>(&render_kernel_stream<false, false, 1024>);
        return reinterpret_cast<const void*>(&render_kernel_stream<false, false, 768>);
    }

    int launch(hipStream_t st, float* out) {
        if (variant == 1) {
            RenderParams p;
            p.width = cfg.width; p.height = cfg.height;
            p.spp = cfg.samples_per_pixel; p.max_depth = cfg.max_depth;
            p.seed = cfg.seed;
            p.cam = cam;
            p.world = scene.dw;
            p.tm = tm;
            p.out = out;
            p.work_counter = work_counter.as<uint32_t>();
            return launch_render(p, variant, st);
        }
        StreamParams p;
        p.width = cfg.width; p.height = cfg.height;
        p.spp = cfg.samples_per_pixel; p.max_depth = cfg.max_depth;
        p.seed = cfg.seed;
        p.cam = cam;
        p.tm = tm;
        p.scene = scene.packed;
        p.world = scene.dw;
        p.scene.n_top = scene.big ? n_top : 0u;
        p.samples = samples.as<float4>();
        p.work_counter = work_counter.as<uint32_t>();
        p.inner_keep = tune[0] ? tune[0] : 1u; p.shade_min = tune[1]; p.leaf_min = tune[2];
        uint32_t n_local_pixels = tm.n_local_tiles * RT_TILE * RT_TILE;
        uint32_t grid = n_cus * stream_blocks_per_cu;
        std::vector<hipEvent_t>& ring = kev[n_renders % RT_TIMES_RING];
        while (ring.size() < (size_t)n_passes * 4u) {
            hipEvent_t e = nullptr;
            HIP_TRY(hipEventCreate(&e));
            ring.push_back(e);
        }
        uint32_t pass = 0;
        for (uint32_t first = 0; first < cfg.samples_per_pixel; first += pass_spp, pass++) {
            p.pass_first_s = first;
            p.pass_spp = std::min(pass_spp, cfg.samples_per_pixel - first);
            p.total = n_local_pixels * p.pass_spp;
            // work-queue granularity: ~32 fetches per wave keep the tail short when a shard is small (multi-GPU)
            uint32_t n_waves = grid * (stream_block / 64u);
            uint32_t chunk = p.total / (n_waves * 32u);
            chunk = std::max(64u, std::min(RT_CHUNK_MAX, chunk & ~63u));
            if (const char* env = std::getenv("RT06_CHUNK")) { int v = std::atoi(env); if (v >= 64 && v <= 1024) chunk = (uint32_t)v & ~63u; }
            p.chunk = chunk;
            // the streaming kernel's waves own their first chunk (chunk w for wave w): the counter starts behind those; the exchange
            // kernel's shader waves draw every chunk from the counter
            const uint64_t first_shared = variant == 5 ? 0ull : (uint64_t)n_waves * chunk;
            HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)work_counter.p, (int)(uint32_t)std::min<uint64_t>(first_shared, 0xF0000000ull), 1, st));
#ifdef RT_PHASE_TIMERS
            DevBuf phase_acc;
            HIP_TRY(phase_acc.alloc((32 + 96 * 16) * sizeof(unsigned long long)));
            HIP_TRY(hipMemsetAsync(phase_acc.p, 0, (32 + 96 * 16) * sizeof(unsigned long long), st));
            p.phase_acc = phase_acc.as<unsigned long long>();
#endif
            const int pb = 0;
            hipEvent_t* ke = ring.data() + (size_t)pass * 4u;
            HIP_TRY(hipEventRecord(ke[0], st));
            {
                const size_t n_pass = (size_t)tm.n_local_tiles * RT_TILE * RT_TILE * pass_spp;   // 16-B records per array
                p.prim_o = primary[pb].as<float4>();
                p.prim_d = primary[pb].as<float4>() + n_pass;
                p.prim_rng = reinterpret_cast<uint4*>(primary[pb].as<float4>() + 2 * n_pass);
            }
            // The generator declares LDS it does not use: more than two resident persistent workgroups leave free on a CU (160 KiB - 2 x
            // ~77 KiB).  Alone on the GPU that changes nothing.  With a SECOND frame in flight on another stream (bench.py --pipeline 2)
            // it keeps the next frame's generator from moving in beside the persistent kernel's main phase (measured harmful) and
            // lets it start exactly when workgroups of the draining frame exit — it fills the tail instead.
            static const uint32_t primary_lds = [] { const char* e = std::getenv("RT06_PRIMARY_LDS"); return e ? (uint32_t)std::atoi(e) : 0u; }();
            for (uint32_t b0 = 0; b0 < tm.n_local_tiles; b0 += 65535u) {   // grid.y = 64-pixel block, at most 65535 per launch
                const uint32_t nb = std::min(65535u, tm.n_local_tiles - b0);
                primary_rays_kernel<<<dim3((64u * p.pass_spp + 255u) / 256u, nb), 256, primary_lds, st>>>(p, b0);
                HIP_TRY(hipGetLastError());
            }

            void* args[] = {&p};
            XchgParams xp;
            if (variant == 5) {
                if (std::getenv("RT06_XCHG")) p.inner_keep = xc.keep;
                xp.s = p;
                xp.n_tracers = xc.n_tracers; xp.n_shards = xc.shards; xp.tq_cap = xc.tq_cap; xp.sq_cap = xc.sq_cap;
                xp.pop_extra = xc.pop_extra;
                xp.swap_min = xc.swap_min; xp.shade_min = xc.shade_min; xp.shade_patience = xc.patience;
                xp.scene_vec4 = xc.scene_vec4; xp.extra_in_lds = xc.extra_in_lds; xp.shader_prio = xc.prio;
                xp.error_flag = xchg_error.as<uint32_t>();
#ifdef RT_PHASE_TIMERS
                xp.xphase_acc = phase_acc.as<unsigned long long>();
#endif
                args[0] = &xp;
            }
            HIP_TRY(hipEventRecord(ke[1], st));
            HIP_TRY(hipLaunchKernel(stream_kernel_ptr(), dim3(grid), dim3(stream_block), args, stream_lds_bytes, st));
            HIP_TRY(hipEventRecord(ke[2], st));
#ifdef RT_PHASE_TIMERS
            {
                static unsigned long long h[32 + 96 * 16];
                HIP_TRY(hipStreamSynchronize(st));
                HIP_TRY(hipMemcpy(h, phase_acc.p, sizeof(h), hipMemcpyDeviceToHost));
                if (const char* hp = std::getenv("RT06_TRACE_HIST")) {   // joint histogram (inner steps x leaf tests) per trace, for tools/sched_model.py
                    if (FILE* f = std::fopen(hp, "w")) {
                        for (int a = 0; a < 96; a++) { for (int b = 0; b < 16; b++) std::fprintf(f, "%llu ", h[32 + a * 16 + b]); std::fprintf(f, "\n"); }
                        std::fclose(f);
                    }
                }
                static const char* names_stream[16] = {"hot inner loop", "irregular loop", "leaf phase", "shade (tail)", "regenerate", "begin trace", "(inner steps)", "loop top",
                                                "schedule check", "shade: miss/sky + hit common", "shade: dielectric prep", "shade: dielectric dir", "shade: on-unit-sphere loop", "shade: metal/lambert/checker", "-", "-"};
                static const char* names_xchg[16] = {"T hot inner loop", "T irregular loop", "T leaf phase", "T exchange", "T idle", "(lanes per hot step)", "(lanes per leaf phase)", "(finished per exchange)",
                                                     "S wait", "S pop", "S shade", "S new samples", "S begin trace", "S push", "(traces per shade round)", "-"};
                const char* const* names = variant == 5 ? names_xchg : names_stream;
                unsigned long long tot = 0;
                for (int i = 0; i < 16; i++) if (!(variant == 5 && (i == 5 || i == 6 || i == 7 || i == 14))) tot += h[i];
                for (int i = 0; i < 16; i++)
                    fprintf(stderr, "[phase] %-16s %6.2f %% of wave time, %12llu visits, %8.1f cycles per visit\n", names[i], 100.0 * h[i] / (double)tot, h[16 + i], h[16 + i] ? (double)h[i] / h[16 + i] : 0.0);
            }
#endif
            uint32_t last = first + p.pass_spp >= cfg.samples_per_pixel ? 1u : 0u;
            resolve_kernel<<<(n_local_pixels + 255) / 256, 256, 0, st>>>(p, running.as<float4>(), out, last);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(ke[3], st));
        }
        n_renders++;
        return RT_OK;
    }
    ~rt_renderer() {
        for (auto& q : kev) for (hipEvent_t e : q) if (e) (void)hipEventDestroy(e);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

extern "C" int rt_renderer_create(const rt_render_config* cfg, const rt_camera* cam, const rt_world_flat* world, rt_renderer** out) {
    if (!cfg || !cam || !world || !out) return rt_fail(RT_ERR_INVALID, "rt_renderer_create: null argument");
    if (cfg->width == 0 || cfg->height == 0 || cfg->samples_per_pixel == 0)
        return rt_fail(RT_ERR_INVALID, "rt_renderer_create: width, height and samples_per_pixel must be > 0");
    if ((uint64_t)cfg->width * cfg->height > 0x7fffffffull) return rt_fail(RT_ERR_INVALID, "rt_renderer_create: image too large");
    if (cfg->world_size == 0 || cfg->rank >= cfg->world_size) return rt_fail(RT_ERR_INVALID, "rt_renderer_create: bad rank %u / world_size %u", cfg->rank, cfg->world_size);
    if (cam->type > RT_CAM_MOTION) return rt_fail(RT_ERR_INVALID, "rt_renderer_create: unknown camera type %u", cam->type);
    int rc = select_device(cfg->device);
    if (rc != RT_OK) return rc;
    rt_renderer* r = new rt_renderer();
    r->cfg = *cfg;
    r->cam = *cam;
    rc = r->scene.upload(world);
    if (rc != RT_OK) { delete r; return rc; }
    r->tm = make_tile_map(cfg->width, cfg->height, cfg->rank, cfg->world_size);
    TileMap& tm = r->tm;
    r->shard_floats = (size_t)tm.n_local_tiles * RT_TILE * RT_TILE * 4;
    size_t fb_floats = tm.direct ? (size_t)cfg->width * cfg->height * 4 : r->shard_floats;
    hipError_t e = r->fb.alloc(fb_floats * sizeof(float));
    if (e == hipSuccess) e = hipMemset(r->fb.p, 0, fb_floats * sizeof(float));
    if (e == hipSuccess) e = r->work_counter.alloc(256);
    if (e == hipSuccess) {
        rc = r->plan();
        if (rc != RT_OK) { delete r; return rc; }
    }
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&r->ev0);
    if (e == hipSuccess) e = hipEventCreate(&r->ev1);
    if (e != hipSuccess) { delete r; return rt_fail(RT_ERR_HIP, "rt_renderer_create: %s", hipGetErrorString(e)); }
    *out = r;
    return RT_OK;
}

extern "C" void rt_renderer_destroy(rt_renderer* r) {
    if (!r) return;
    (void)hipSetDevice(r->cfg.device);
    delete r;
}

extern "C" int rt_renderer_render_async(rt_renderer* r, void* hip_stream, float* d_out) {
    if (!r) return rt_fail(RT_ERR_INVALID, "rt_renderer_render_async: null renderer");
    HIP_TRY(hipSetDevice(r->cfg.device));
    hipStream_t st = (hipStream_t)hip_stream;  // NULL is the HIP null stream, as for any HIP launch
    HIP_TRY(hipEventRecord(r->ev0, st));
    int rc = r->launch(st, d_out ? d_out : r->fb.as<float>());
    if (rc != RT_OK) return rc;
    HIP_TRY(hipEventRecord(r->ev1, st));
    r->timed = true;
    return RT_OK;
}

extern "C" int rt_renderer_render(rt_renderer* r) {
    if (!r) return rt_fail(RT_ERR_INVALID, "rt_renderer_render: null renderer");
    int rc = rt_renderer_render_async(r, r->stream, nullptr);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipStreamSynchronize(r->stream));
    rc = check_xchg_error(r->xchg_error);
    if (rc != RT_OK) return rc;
    return check_traversal_overflow(r->scene);
}

extern "C" int rt_renderer_last_kernel_ms(rt_renderer* r, float* out_ms) {
    if (!r || !out_ms) return rt_fail(RT_ERR_INVALID, "rt_renderer_last_kernel_ms: null argument");
    if (!r->timed) return rt_fail(RT_ERR_INVALID, "rt_renderer_last_kernel_ms: nothing rendered yet");
    HIP_TRY(hipSetDevice(r->cfg.device));
    HIP_TRY(hipEventSynchronize(r->ev1));
    HIP_TRY(hipEventElapsedTime(out_ms, r->ev0, r->ev1));
    return RT_OK;
}

extern "C" int rt_renderer_kernel_times(rt_renderer* r, uint32_t renders_back, float out_ms[3]) {
    if (!r || !out_ms) return rt_fail(RT_ERR_INVALID, "rt_renderer_kernel_times: null argument");
    if (r->variant < 2) return rt_fail(RT_ERR_INVALID, "rt_renderer_kernel_times: the baseline kernel (variant 1) is one launch; use rt_renderer_last_kernel_ms");
    if (renders_back >= rt_renderer::RT_TIMES_RING || renders_back >= r->n_renders)
        return rt_fail(RT_ERR_INVALID, "rt_renderer_kernel_times: render %u calls back is not recorded (%llu rendered, ring of %u)", renders_back,
                       (unsigned long long)r->n_renders, rt_renderer::RT_TIMES_RING);
    HIP_TRY(hipSetDevice(r->cfg.device));
    const std::vector<hipEvent_t>& ring = r->kev[(r->n_renders - 1 - renders_back) % rt_renderer::RT_TIMES_RING];
    HIP_TRY(hipEventSynchronize(ring[(size_t)r->n_passes * 4u - 1u]));
    for (int k = 0; k < 3; k++) out_ms[k] = 0.0f;
    for (uint32_t pass = 0; pass < r->n_passes; pass++)   // a render is n_passes launches of each kernel: the SUM is the render's time in it
        for (int k = 0; k < 3; k++) {
            float ms = 0.0f;
            HIP_TRY(hipEventElapsedTime(&ms, ring[pass * 4u + k], ring[pass * 4u + k + 1]));
            out_ms[k] += ms;
        }
    return RT_OK;
}

extern "C" int rt_renderer_pass_info(rt_renderer* r, uint64_t out[4]) {
    if (!r || !out) return rt_fail(RT_ERR_INVALID, "rt_renderer_pass_info: null argument");
    out[0] = r->variant >= 2 ? r->n_passes : 1u;
    out[1] = r->variant >= 2 ? r->pass_spp : r->cfg.samples_per_pixel;
    out[2] = r->variant >= 2 ? rt_renderer::SAMPLE_BYTES + rt_renderer::PRIMARY_BYTES : 0u;
    out[3] = r->samples.bytes + r->primary[0].bytes + r->running.bytes;
    return RT_OK;
}

extern "C" int rt_renderer_kernel_info(rt_renderer* r, uint32_t out[4]) {
    if (!r || !out) return rt_fail(RT_ERR_INVALID, "rt_renderer_kernel_info: null argument");
    out[0] = r->tol ? 5u + r->tol : r->variant;
    out[1] = (r->variant >= 2 && !r->scene.big) ? 1u : 0u;
    out[2] = r->variant >= 2 ? r->stream_block : 64u;
    out[3] = r->variant >= 2 ? r->stream_blocks_per_cu : 0u;
    return RT_OK;
}

extern "C" int rt_renderer_download(rt_renderer* r, float* host_rgba, size_t n_floats) {
    if (!r || !host_rgba) return rt_fail(RT_ERR_INVALID, "rt_renderer_download: null argument");
    if (r->cfg.world_size != 1) return rt_fail(RT_ERR_INVALID, "rt_renderer_download: renderer holds one shard of %u; gather and rt_renderer_assemble first", r->cfg.world_size);
    size_t need = (size_t)r->cfg.width * r->cfg.height * 4;
    if (n_floats != need) return rt_fail(RT_ERR_INVALID, "rt_renderer_download: buffer holds %zu floats, image needs %zu", n_floats, need);
    HIP_TRY(hipSetDevice(r->cfg.device));
    // the last render may have been launched on a caller's stream (rt_renderer_render_async): its end event orders the copy
    if (r->timed) HIP_TRY(hipEventSynchronize(r->ev1));
    HIP_TRY(hipStreamSynchronize(r->stream));
    HIP_TRY(hipMemcpy(host_rgba, r->fb.p, need * sizeof(float), hipMemcpyDeviceToHost));
    int rc = check_xchg_error(r->xchg_error);
    if (rc != RT_OK) return rc;
    return check_traversal_overflow(r->scene);
}

extern "C" int rt_renderer_shard_floats(const rt_renderer* r, size_t* out) {
    if (!r || !out) return rt_fail(RT_ERR_INVALID, "rt_renderer_shard_floats: null argument");
    *out = r->shard_floats;
    return RT_OK;
}

// de-interleave the gathered shards (rank-major, tile-major inside a shard) into the row-major image
__global__ void assemble_kernel(const float4* __restrict__ gathered, float4* __restrict__ image, TileMap tm, uint32_t shard_pixels) {
    uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= tm.width * tm.height) return;
    uint32_t x = gid % tm.width, y = gid / tm.width;
    uint32_t gt = (y / RT_TILE) * tm.tiles_x + (x / RT_TILE);
    uint32_t rank = gt % tm.world_size, tl = gt / tm.world_size;
    uint32_t p = (y % RT_TILE) * RT_TILE + (x % RT_TILE);
    image[gid] = gathered[(size_t)rank * shard_pixels + (size_t)tl * (RT_TILE * RT_TILE) + p];
}

extern "C" int rt_renderer_assemble(rt_renderer* r, const float* d_gathered, float* d_image, void* hip_stream) {
    if (!r || !d_gathered || !d_image) return rt_fail(RT_ERR_INVALID, "rt_renderer_assemble: null argument");
    HIP_TRY(hipSetDevice(r->cfg.device));
    hipStream_t st = (hipStream_t)hip_stream;
    uint32_t n = r->cfg.width * r->cfg.height;
    assemble_kernel<<<(n + 255) / 256, 256, 0, st>>>((const float4*)d_gathered, (float4*)d_image, r->tm,
                                                      (uint32_t)(r->shard_floats / 4));
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

// ---------------------------------------------------------------------------------------------
// Multi-GPU renderer: ONE host process drives the N GPUs of a node (SURVEY.md §5 last row, §8e).  Rank i = device i renders
// the tiles t with t % N == i into its compact shard; at frame end ONE grouped RCCL exchange moves the shards to device 0
// over xGMI (every peer has its own link to the root, so the N - 1 transfers run side by side), and assemble_kernel
// de-interleaves them into the row-major image there.  Nothing else is communicated: the scene is replicated (tens of KB)
// and the RNG is keyed by global pixel and sample, so the image has the same bits for every N.
// RCCL is bound at first use (dlopen of librccl.so.1: ncclCommInitAll, ncclGroupStart/End, ncclSend, ncclRecv,
// ncclCommDestroy, ncclGetErrorString) so that single-GPU callers do not map the 570-MB collective library.
// ---------------------------------------------------------------------------------------------
namespace {
struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
RcclApi g_rccl;
std::once_flag g_rccl_once;
int g_rccl_rc = RT_OK;
std::string g_rccl_error;

int rccl_bind_once() {
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return rt_fail(RT_ERR_HIP, "multi-GPU rendering needs RCCL: %s", dlerror());
    RcclApi a;
    a.handle = h;
#define RT_RCCL_SYM(field, name)                                                                    \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, name));                                  \
    if (!a.field) return rt_fail(RT_ERR_HIP, "librccl.so.1 has no symbol %s", name)
    RT_RCCL_SYM(CommInitAll, "ncclCommInitAll");
    RT_RCCL_SYM(CommDestroy, "ncclCommDestroy");
    RT_RCCL_SYM(GroupStart, "ncclGroupStart");
    RT_RCCL_SYM(GroupEnd, "ncclGroupEnd");
    RT_RCCL_SYM(Send, "ncclSend");
    RT_RCCL_SYM(Recv, "ncclRecv");
    RT_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef RT_RCCL_SYM
    g_rccl = a;
    return RT_OK;
}
// bound once per process, whichever thread creates the first multi-GPU renderer
int rccl_bind() {
    std::call_once(g_rccl_once, [] {
        g_rccl_rc = rccl_bind_once();
        if (g_rccl_rc != RT_OK) g_rccl_error = rt_last_error();
    });
    return g_rccl_rc == RT_OK ? RT_OK : rt_fail(g_rccl_rc, "%s", g_rccl_error.c_str());
}
}  // namespace

#define RCCL_TRY(expr)                                                                                                  \
    do {                                                                                                                \
        ncclResult_t _r = (expr);                                                                                       \
        if (_r != ncclSuccess) return rt_fail(RT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__); \
    } while (0)

// How the shards travel to devices[0] at frame end.
//   RT_TRANSPORT_RCCL   (default): one grouped ncclSend / ncclRecv exchange over xGMI, one rank per GPU.
//   RT_TRANSPORT_MEMCPY (RT06_MULTI_TRANSPORT=memcpy; tests and single-GPU boxes): hipMemcpyAsync on the ranks' own streams, ordered
//       by events.  It lifts the one-rank-per-device rule, so N ranks can share ONE GPU and the whole N > 1 branch — shard offsets,
//       stream ordering, assemble_kernel, download — runs where RCCL would refuse (it does not accept two ranks on one device).
enum : uint32_t { RT_TRANSPORT_RCCL = 0, RT_TRANSPORT_MEMCPY = 1 };

struct rt_multi_renderer {
    uint32_t width = 0, height = 0;
    uint32_t transport = RT_TRANSPORT_RCCL;
    std::vector<int> devices;
    std::vector<rt_renderer*> parts;     // parts[i]: rank i of N on devices[i]
    std::vector<ncclComm_t> comms;
    std::vector<hipEvent_t> ev_part;     // per rank, on its device: its render is enqueued / (memcpy transport) its shard has been copied
    DevBuf gathered, image;              // on devices[0]: N shards back to back; the assembled row-major frame
    hipEvent_t ev_rendered = nullptr, ev_done = nullptr;   // on devices[0]'s stream: every rank has rendered / after the assembly
    float last_total_ms = 0.0f;
    bool rendered = false;
    ~rt_multi_renderer() {
        for (ncclComm_t c : comms) if (c) (void)g_rccl.CommDestroy(c);
        for (size_t i = 0; i < ev_part.size(); i++) {
            if (!ev_part[i]) continue;
            (void)hipSetDevice(devices[i]);
            (void)hipEventDestroy(ev_part[i]);
        }
        for (rt_renderer* r : parts) rt_renderer_destroy(r);
        if (!devices.empty()) (void)hipSetDevice(devices[0]);
        if (ev_rendered) (void)hipEventDestroy(ev_rendered);
        if (ev_done) (void)hipEventDestroy(ev_done);
    }
};

extern "C" int rt_multi_renderer_create(const rt_render_config* cfg, const rt_camera* cam, const rt_world_flat* world, uint32_t n_gpus,
                                        const int32_t* devices, rt_multi_renderer** out) {
    if (!cfg || !cam || !world || !out) return rt_fail(RT_ERR_INVALID, "rt_multi_renderer_create: null argument");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0)
        return rt_fail(RT_ERR_NO_DEVICE, "no HIP device available: the HIP path is required, there is no CPU fallback");
    uint32_t transport = RT_TRANSPORT_RCCL;
    if (const char* env = std::getenv("RT06_MULTI_TRANSPORT")) {
        if (std::strcmp(env, "memcpy") == 0) transport = RT_TRANSPORT_MEMCPY;
        else if (std::strcmp(env, "rccl") != 0) return rt_fail(RT_ERR_INVALID, "RT06_MULTI_TRANSPORT=%s: expected rccl or memcpy", env);
    }
    if (n_gpus == 0 || n_gpus > 64) return rt_fail(RT_ERR_INVALID, "rt_multi_renderer_create: %u ranks asked for", n_gpus);
    if (transport == RT_TRANSPORT_RCCL && (int)n_gpus > n_dev) return rt_fail(RT_ERR_INVALID, "rt_multi_renderer_create: %u GPUs asked for, %d present", n_gpus, n_dev);
    std::vector<int> devs(n_gpus);
    for (uint32_t i = 0; i < n_gpus; i++) {
        devs[i] = devices ? devices[i] : (transport == RT_TRANSPORT_MEMCPY ? (int)(i % (uint32_t)n_dev) : (int)i);
        if (devs[i] < 0 || devs[i] >= n_dev) return rt_fail(RT_ERR_INVALID, "rt_multi_renderer_create: device %d out of range", devs[i]);
        for (uint32_t j = 0; j < i && transport == RT_TRANSPORT_RCCL; j++)
            if (devs[j] == devs[i]) return rt_fail(RT_ERR_INVALID, "rt_multi_renderer_create: device %d listed twice (one rank per GPU)", devs[i]);
    }
    int rc = transport == RT_TRANSPORT_RCCL ? rccl_bind() : RT_OK;
    if (rc != RT_OK) return rc;
    std::unique_ptr<rt_multi_renderer> m(new rt_multi_renderer());
    m->width = cfg->width; m->height = cfg->height;
    m->transport = transport;
    m->devices = devs;
    for (uint32_t i = 0; i < n_gpus; i++) {
        rt_render_config c = *cfg;
        c.device = devs[i]; c.rank = i; c.world_size = n_gpus;
        rt_renderer* r = nullptr;
        rc = rt_renderer_create(&c, cam, world, &r);
        if (rc != RT_OK) return rc;
        m->parts.push_back(r);
    }
    if (transport == RT_TRANSPORT_RCCL) {
        m->comms.assign(n_gpus, nullptr);
        RCCL_TRY(g_rccl.CommInitAll(m->comms.data(), (int)n_gpus, devs.data()));
    }
    m->ev_part.assign(n_gpus, nullptr);
    for (uint32_t i = 0; i < n_gpus; i++) {
        HIP_TRY(hipSetDevice(devs[i]));
        HIP_TRY(hipEventCreateWithFlags(&m->ev_part[i], hipEventDisableTiming));
    }
    HIP_TRY(hipSetDevice(devs[0]));
    const size_t image_floats = (size_t)cfg->width * cfg->height * 4;
    HIP_TRY(m->image.alloc(image_floats * sizeof(float)));
    if (n_gpus > 1) HIP_TRY(m->gathered.alloc(m->parts[0]->shard_floats * n_gpus * sizeof(float)));
    HIP_TRY(hipEventCreate(&m->ev_rendered));
    HIP_TRY(hipEventCreate(&m->ev_done));
    *out = m.release();
    return RT_OK;
}

extern "C" void rt_multi_renderer_destroy(rt_multi_renderer* m) { delete m; }

// a failure between the launches and the final synchronisation must not leave work in flight on the ranks' streams
static int multi_fail_drain(rt_multi_renderer* m, int rc) {
    const std::string msg = rt_last_error();   // the drains below may overwrite the message of the failure we report
    for (size_t i = 0; i < m->parts.size(); i++)
        if (hipSetDevice(m->devices[i]) == hipSuccess) (void)hipStreamSynchronize(m->parts[i]->stream);
    return rt_fail(rc, "%s", msg.c_str());
}

extern "C" int rt_multi_renderer_render(rt_multi_renderer* m) {
    if (!m) return rt_fail(RT_ERR_INVALID, "rt_multi_renderer_render: null renderer");
    const uint32_t n = (uint32_t)m->parts.size();
    const auto t0 = std::chrono::steady_clock::now();
    hipStream_t s0 = m->parts[0]->stream;
    for (uint32_t i = 0; i < n; i++) {   // every GPU renders its tiles; the launches are asynchronous, so the N kernels run side by side
        int rc = rt_renderer_render_async(m->parts[i], m->parts[i]->stream, nullptr);
        if (rc != RT_OK) return multi_fail_drain(m, rc);
        HIP_TRY(hipEventRecord(m->ev_part[i], m->parts[i]->stream));
    }
    // stream 0 waits for EVERY rank's render before the exchange timer starts: times()[2] is then exchange + assembly, not the
    // slowest rank's tail (the receive would otherwise absorb the imbalance of the frame)
    HIP_TRY(hipSetDevice(m->devices[0]));
    for (uint32_t i = 1; i < n; i++) HIP_TRY(hipStreamWaitEvent(s0, m->ev_part[i], 0));
    HIP_TRY(hipEventRecord(m->ev_rendered, s0));
    // the single frame-end exchange: rank i sends its shard to rank 0 (rank 0 to itself), rank 0 receives N shards in rank order.
    // With one GPU this is the degenerate self-exchange of the whole row-major frame.
    const size_t count = n > 1 ? m->parts[0]->shard_floats : (size_t)m->width * m->height * 4;
    float* dst = n > 1 ? m->gathered.as<float>() : m->image.as<float>();
    if (m->transport == RT_TRANSPORT_RCCL) {
        // an error inside the group still CLOSES the group (an open group makes the process's next collective call hang)
        ncclResult_t first = g_rccl.GroupStart();
        const char* what = "ncclGroupStart";
        if (first == ncclSuccess) {
            for (uint32_t i = 0; i < n && first == ncclSuccess; i++) {
                first = g_rccl.Send(m->parts[i]->fb.p, count, ncclFloat, 0, m->comms[i], m->parts[i]->stream);
                what = "ncclSend";
            }
            for (uint32_t i = 0; i < n && first == ncclSuccess; i++) {
                first = g_rccl.Recv(dst + (size_t)i * count, count, ncclFloat, (int)i, m->comms[0], s0);
                what = "ncclRecv";
            }
            const ncclResult_t end = g_rccl.GroupEnd();
            if (first == ncclSuccess && end != ncclSuccess) { first = end; what = "ncclGroupEnd"; }
        }
        if (first != ncclSuccess) {
            (void)rt_fail(RT_ERR_HIP, "%s failed in the frame-end exchange: %s", what, g_rccl.GetErrorString(first));
            return multi_fail_drain(m, RT_ERR_HIP);
        }
    } else {
        for (uint32_t i = 0; i < n; i++) {
            HIP_TRY(hipSetDevice(m->devices[i]));
            HIP_TRY(hipMemcpyAsync(dst + (size_t)i * count, m->parts[i]->fb.p, count * sizeof(float), hipMemcpyDeviceToDevice, m->parts[i]->stream));
            HIP_TRY(hipEventRecord(m->ev_part[i], m->parts[i]->stream));
        }
        HIP_TRY(hipSetDevice(m->devices[0]));
        for (uint32_t i = 1; i < n; i++) HIP_TRY(hipStreamWaitEvent(s0, m->ev_part[i], 0));
    }
    HIP_TRY(hipSetDevice(m->devices[0]));
    if (n > 1) {
        int rc = rt_renderer_assemble(m->parts[0], m->gathered.as<float>(), m->image.as<float>(), s0);
        if (rc != RT_OK) return multi_fail_drain(m, rc);
    }
    HIP_TRY(hipEventRecord(m->ev_done, s0));
    for (uint32_t i = 0; i < n; i++) {
        HIP_TRY(hipSetDevice(m->devices[i]));
        HIP_TRY(hipStreamSynchronize(m->parts[i]->stream));
    }
    for (uint32_t i = 0; i < n; i++) {   // RT_TRAVERSAL_QUEUE / _WIDE4 worlds: an overflow of the 32 entries is an error here too
        HIP_TRY(hipSetDevice(m->devices[i]));
        int rc = check_xchg_error(m->parts[i]->xchg_error);
        if (rc == RT_OK) rc = check_traversal_overflow(m->parts[i]->scene);
        if (rc != RT_OK) return rc;
    }
    m->last_total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    m->rendered = true;
    return RT_OK;
}

extern "C" int rt_multi_renderer_download(rt_multi_renderer* m, float* host_rgba, size_t n_floats) {
    if (!m || !host_rgba) return rt_fail(RT_ERR_INVALID, "rt_multi_renderer_download: null argument");
    const size_t need = (size_t)m->width * m->height * 4;
    if (n_floats != need) return rt_fail(RT_ERR_INVALID, "rt_multi_renderer_download: buffer holds %zu floats, image needs %zu", n_floats, need);
    HIP_TRY(hipSetDevice(m->devices[0]));
    HIP_TRY(hipStreamSynchronize(m->parts[0]->stream));
    HIP_TRY(hipMemcpy(host_rgba, m->image.p, need * sizeof(float), hipMemcpyDeviceToHost));
    return RT_OK;
}

extern "C" int rt_multi_renderer_times(rt_multi_renderer* m, float out_ms[3]) {
    if (!m || !out_ms) return rt_fail(RT_ERR_INVALID, "rt_multi_renderer_times: null argument");
    if (!m->rendered) return rt_fail(RT_ERR_INVALID, "rt_multi_renderer_times: nothing rendered yet");
    out_ms[0] = m->last_total_ms;
    float worst = 0.0f;
    for (rt_renderer* r : m->parts) {
        float ms = 0.0f;
        int rc = rt_renderer_last_kernel_ms(r, &ms);
        if (rc != RT_OK) return rc;
        worst = std::max(worst, ms);
    }
    out_ms[1] = worst;
    HIP_TRY(hipSetDevice(m->devices[0]));
    HIP_TRY(hipEventElapsedTime(out_ms + 2, m->ev_rendered, m->ev_done));
    return RT_OK;
}

extern "C" int rt_multi_renderer_gpus(const rt_multi_renderer* m, uint32_t* out) {
    if (!m || !out) return rt_fail(RT_ERR_INVALID, "rt_multi_renderer_gpus: null argument");
    *out = (uint32_t)m->parts.size();
    return RT_OK;
}

// ---------------------------------------------------------------------------------------------
// probes
// ---------------------------------------------------------------------------------------------
__global__ void probe_aabb_kernel(size_t n, const float* boxes, const float* rays, const float* maxd, int32_t* hit, float* dist) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ray r;
    r.o = ld3(rays + 6 * i); r.d = ld3(rays + 6 * i + 3); r.time = 0.0f;
    float d = 0.0f;
    hit[i] = aabb_intersects(ld3(boxes + 6 * i), ld3(boxes + 6 * i + 3), r, maxd[i], d) ? 1 : 0;
    dist[i] = d;
}
__global__ void probe_sphere_kernel(size_t n, const float* rays, const float* spheres, float* out_t) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ray r;
    r.o = ld3(rays + 6 * i); r.d = ld3(rays + 6 * i + 3); r.time = 0.0f;
    out_t[i] = sphere_closest_intersection(r, ld3(spheres + 4 * i), spheres[4 * i + 3]);
}
__global__ void probe_trace_kernel(DeviceWorld w, size_t n, const float* rays, int32_t* hit, float* t, int32_t* prim, float* normal) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ray r;
    r.o = ld3(rays + 7 * i); r.d = ld3(rays + 7 * i + 3); r.time = rays[7 * i + 6];
    HitRec rec;
    rec.distance = RT_MISS_DIST; rec.normal = mk3(0.0f); rec.prim = -1; rec.mat = 0;
    Rng g;
    g.init(0u, (uint32_t)i, 0u, 0x7ACEu);  // only a constant medium draws from it (same key as the oracle's probe)
    hit[i] = world_closest_intersection(w, r, rec, &g) ? 1 : 0;
    t[i] = rec.distance; prim[i] = rec.prim;
    st3(normal + 3 * i, rec.normal);
}
__global__ void probe_scatter_kernel(uint64_t seed, size_t n, const rt_material* mats, const float* rays, const float* dist,
                                     const float* normals, const uint32_t* keys, int32_t* scattered, float* out_rays,
                                     float* atten, uint32_t* draws) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ray in;
    in.o = ld3(rays + 7 * i); in.d = ld3(rays + 7 * i + 3); in.time = rays[7 * i + 6];
    HitRec rec;
    rec.distance = dist[i]; rec.normal = ld3(normals + 3 * i); rec.prim = 0; rec.mat = 0;
    Rng g;
    g.init(seed, keys[2 * i], keys[2 * i + 1], RT_STREAM_RENDER);
    Ray out;
    out.o = mk3(0.0f); out.d = mk3(0.0f); out.time = 0.0f;
    f3 att = mk3(0.0f);
    scattered[i] = material_scatter(mats[i], in, rec, g, out, att) ? 1 : 0;
    st3(out_rays + 7 * i, out.o); st3(out_rays + 7 * i + 3, out.d); out_rays[7 * i + 6] = out.time;
    st3(atten + 3 * i, att);
    draws[i] = g.draws;
}
__global__ void probe_camera_kernel(uint64_t seed, rt_camera cam, size_t n, const float* st, const uint32_t* keys, float* out_rays, uint32_t* draws) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Rng g;
    g.init(seed, keys[2 * i], keys[2 * i + 1], RT_STREAM_RENDER);
    Ray r = camera_sample_ray(cam, st[2 * i], st[2 * i + 1], g);
    st3(out_rays + 7 * i, r.o); st3(out_rays + 7 * i + 3, r.d); out_rays[7 * i + 6] = r.time;
    draws[i] = g.draws;
}
__global__ void probe_radiance_kernel(DeviceWorld w, rt_camera cam, uint32_t width, uint32_t height, uint32_t max_depth,
                                      uint64_t seed, size_t n, const uint32_t* keys, float* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f3 rad = one_sample(w, cam, width, height, max_depth, seed, keys[2 * i], keys[2 * i + 1]);
    st3(out + 3 * i, rad);
}
// The reference's one gtest computes, per pixel, the index of the nearest sphere by brute force (google_testing/test.cpp:112-135,
// host twin :87-106).  Here: one work-item per pixel of a flat index space, the sphere table staged through the LDS in slabs of
// 256 so that the 64 lanes of a wave read each sphere as a broadcast; NDC = i / (extent - 1) * 2 - 1 is that test's convention
// (test.cpp:118-119), not the renderer's pixel-centre one.
__global__ __launch_bounds__(256) void probe_sphere_index_kernel(const float4* __restrict__ spheres, uint32_t n_spheres, rt_camera cam,
                                                                 uint32_t width, uint32_t height, int32_t* __restrict__ nearest) {
    __shared__ float4 slab[256];
    const uint32_t pixel = blockIdx.x * 256u + threadIdx.x;
    const bool live = pixel < width * height;
    const uint32_t px = live ? pixel % width : 0u, py = live ? pixel / width : 0u;
    Ray ray;
    ray.o = mk3(cam.o[0], cam.o[1], cam.o[2]);
    const float s = (float)px / ((float)width - 1.0f) * 2 - 1, t = (float)py / ((float)height - 1.0f) * 2 - 1;
    ray.d = mk3(cam.w[0], cam.w[1], cam.w[2]) + mk3(cam.u[0], cam.u[1], cam.u[2]) * s + mk3(cam.v[0], cam.v[1], cam.v[2]) * t;
    ray.time = 0.0f;
    float nearest_t = RT_MISS_DIST;
    int32_t winner = -1;
    for (uint32_t base = 0; base < n_spheres; base += 256u) {
        const uint32_t count = min(256u, n_spheres - base);
        __syncthreads();
        if (threadIdx.x < count) slab[threadIdx.x] = spheres[base + threadIdx.x];
        __syncthreads();
        for (uint32_t k = 0; k < count; k++) {
            const float4 sp = slab[k];
            const float tk = sphere_closest_intersection(ray, mk3(sp.x, sp.y, sp.z), sp.w);
            if (tk < nearest_t) { nearest_t = tk; winner = (int32_t)(base + k); }   // strict: the first of equal distances wins
        }
    }
    if (live) nearest[pixel] = winner;
}
__global__ void probe_rng_kernel(uint64_t seed, size_t n, const uint32_t* keys, uint32_t n_draws, float* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Rng g;
    g.init(seed, keys[2 * i], keys[2 * i + 1], RT_STREAM_RENDER);
    for (uint32_t k = 0; k < n_draws; k++) out[i * n_draws + k] = g.next();
}

#define PROBE_GRID(n) dim3((unsigned)(((n) + 127) / 128)), dim3(128)
#define UP(buf, src, bytes) HIP_TRY((buf).upload((src), (bytes)))
#define DOWN(dst, buf, bytes) HIP_TRY(hipMemcpy((dst), (buf).p, (bytes), hipMemcpyDeviceToHost))
#define FINISH()                     \
    HIP_TRY(hipGetLastError());      \
    HIP_TRY(hipDeviceSynchronize())

extern "C" int rt_probe_aabb(int device, size_t n, const float* boxes, const float* rays, const float* max_dist, int32_t* out_hit, float* out_dist) {
    if (!boxes || !rays || !max_dist || !out_hit || !out_dist) return rt_fail(RT_ERR_INVALID, "rt_probe_aabb: null argument");
    if (n == 0) return RT_OK;
    int rc = select_device(device);
    if (rc != RT_OK) return rc;
    DevBuf b, r, m, h, d;
    UP(b, boxes, n * 24); UP(r, rays, n * 24); UP(m, max_dist, n * 4);
    HIP_TRY(h.alloc(n * 4)); HIP_TRY(d.alloc(n * 4));
    probe_aabb_kernel<<<PROBE_GRID(n)>>>(n, b.as<float>(), r.as<float>(), m.as<float>(), h.as<int32_t>(), d.as<float>());
    FINISH();
    DOWN(out_hit, h, n * 4); DOWN(out_dist, d, n * 4);
    return RT_OK;
}
extern "C" int rt_probe_sphere(int device, size_t n, const float* rays, const float* spheres, float* out_t) {
    if (!rays || !spheres || !out_t) return rt_fail(RT_ERR_INVALID, "rt_probe_sphere: null argument");
    if (n == 0) return RT_OK;
    int rc = select_device(device);
    if (rc != RT_OK) return rc;
    DevBuf r, s, t;
    UP(r, rays, n * 24); UP(s, spheres, n * 16);
    HIP_TRY(t.alloc(n * 4));
    probe_sphere_kernel<<<PROBE_GRID(n)>>>(n, r.as<float>(), s.as<float>(), t.as<float>());
    FINISH();
    DOWN(out_t, t, n * 4);
    return RT_OK;
}
extern "C" int rt_probe_trace(int device, const rt_world_flat* world, size_t n, const float* rays, int32_t* out_hit, float* out_t,
                              int32_t* out_prim, float* out_normal) {
    if (!rays || !out_hit || !out_t || !out_prim || !out_normal) return rt_fail(RT_ERR_INVALID, "rt_probe_trace: null argument");
    int rc = select_device(device);
    if (rc != RT_OK) return rc;
    DeviceScene sc;
    rc = sc.upload(world);
    if (rc != RT_OK) return rc;
    if (n == 0) return RT_OK;
    DevBuf r, h, t, p, nn;
    UP(r, rays, n * 28);
    HIP_TRY(h.alloc(n * 4)); HIP_TRY(t.alloc(n * 4)); HIP_TRY(p.alloc(n * 4)); HIP_TRY(nn.alloc(n * 12));
    probe_trace_kernel<<<PROBE_GRID(n)>>>(sc.dw, n, r.as<float>(), h.as<int32_t>(), t.as<float>(), p.as<int32_t>(), nn.as<float>());
    FINISH();
    DOWN(out_hit, h, n * 4); DOWN(out_t, t, n * 4); DOWN(out_prim, p, n * 4); DOWN(out_normal, nn, n * 12);
    return check_traversal_overflow(sc);
}
extern "C" int rt_probe_scatter(int device, uint64_t seed, size_t n, const rt_material* mats, const float* rays, const float* dist,
                                const float* normals, const uint32_t* keys, int32_t* out_scattered, float* out_rays, float* out_atten,
                                uint32_t* out_draws) {
    if (!mats || !rays || !dist || !normals || !keys || !out_scattered || !out_rays || !out_atten || !out_draws)
        return rt_fail(RT_ERR_INVALID, "rt_probe_scatter: null argument");
    if (n == 0) return RT_OK;
    for (size_t i = 0; i < n; i++)
        if (mats[i].type > RT_MAT_ISOTROPIC) return rt_fail(RT_ERR_INVALID, "rt_probe_scatter: case %zu: unknown material type", i);
    int rc = select_device(device);
    if (rc != RT_OK) return rc;
    DevBuf m, r, d, nn, k, s, orr, a, dr;
    UP(m, mats, n * sizeof(rt_material)); UP(r, rays, n * 28); UP(d, dist, n * 4); UP(nn, normals, n * 12); UP(k, keys, n * 8);
    HIP_TRY(s.alloc(n * 4)); HIP_TRY(orr.alloc(n * 28)); HIP_TRY(a.alloc(n * 12)); HIP_TRY(dr.alloc(n * 4));
    probe_scatter_kernel<<<PROBE_GRID(n)>>>(seed, n, m.as<rt_material>(), r.as<float>(), d.as<float>(), nn.as<float>(), k.as<uint32_t>(),
                                            s.as<int32_t>(), orr.as<float>(), a.as<float>(), dr.as<uint32_t>());
    FINISH();
    DOWN(out_scattered, s, n * 4); DOWN(out_rays, orr, n * 28); DOWN(out_atten, a, n * 12); DOWN(out_draws, dr, n * 4);
    return RT_OK;
}
extern "C" int rt_probe_camera(int device, uint64_t seed, const rt_camera* cam, size_t n, const float* st, const uint32_t* keys,
                               float* out_rays, uint32_t* out_draws) {
    if (!cam || !st || !keys || !out_rays || !out_draws) return rt_fail(RT_ERR_INVALID, "rt_probe_camera: null argument");
    if (cam->type > RT_CAM_MOTION) return rt_fail(RT_ERR_INVALID, "rt_probe_camera: unknown camera type");
    if (n == 0) return RT_OK;
    int rc = select_device(device);
    if (rc != RT_OK) return rc;
    DevBuf s, k, r, d;
    UP(s, st, n * 8); UP(k, keys, n * 8);
    HIP_TRY(r.alloc(n * 28)); HIP_TRY(d.alloc(n * 4));
    probe_camera_kernel<<<PROBE_GRID(n)>>>(seed, *cam, n, s.as<float>(), k.as<uint32_t>(), r.as<float>(), d.as<uint32_t>());
    FINISH();
    DOWN(out_rays, r, n * 28); DOWN(out_draws, d, n * 4);
    return RT_OK;
}
extern "C" int rt_probe_radiance(const rt_render_config* cfg, const rt_camera* cam, const rt_world_flat* world, size_t n,
                                 const uint32_t* keys, float* out_radiance) {
    if (!cfg || !cam || !keys || !out_radiance) return rt_fail(RT_ERR_INVALID, "rt_probe_radiance: null argument");
    if (cfg->width == 0 || cfg->height == 0) return rt_fail(RT_ERR_INVALID, "rt_probe_radiance: empty image");
    if (cam->type > RT_CAM_MOTION) return rt_fail(RT_ERR_INVALID, "rt_probe_radiance: unknown camera type");
    for (size_t i = 0; i < n; i++)
        if (keys[2 * i] >= cfg->width * cfg->height) return rt_fail(RT_ERR_INVALID, "rt_probe_radiance: key %zu: pixel out of range", i);
    int rc = select_device(cfg->device);
    if (rc != RT_OK) return rc;
    DeviceScene sc;
    rc = sc.upload(world);
    if (rc != RT_OK) return rc;
    if (n == 0) return RT_OK;
    DevBuf k, o;
    UP(k, keys, n * 8);
    HIP_TRY(o.alloc(n * 12));
    probe_radiance_kernel<<<PROBE_GRID(n)>>>(sc.dw, *cam, cfg->width, cfg->height, cfg->max_depth, cfg->seed, n, k.as<uint32_t>(), o.as<float>());
    FINISH();
    DOWN(out_radiance, o, n * 12);
    return check_traversal_overflow(sc);
}
extern "C" int rt_probe_sphere_index(int device, const rt_camera* cam, uint32_t width, uint32_t height, size_t n_spheres,
                                     const float* spheres, int32_t* out_index) {
    if (!cam || !spheres || !out_index) return rt_fail(RT_ERR_INVALID, "rt_probe_sphere_index: null argument");
    if (width == 0 || height == 0) return RT_OK;
    int rc = select_device(device);
    if (rc != RT_OK) return rc;
    DevBuf s, o;
    UP(s, spheres, n_spheres * 16);
    HIP_TRY(o.alloc((size_t)width * height * 4));
    if (n_spheres > 0x7fffffffull || (uint64_t)width * height > 0xffffff00ull) return rt_fail(RT_ERR_INVALID, "rt_probe_sphere_index: too large");
    probe_sphere_index_kernel<<<(width * height + 255u) / 256u, 256>>>(s.as<float4>(), (uint32_t)n_spheres, *cam, width, height, o.as<int32_t>());
    FINISH();
    DOWN(out_index, o, (size_t)width * height * 4);
    return RT_OK;
}
extern "C" int rt_probe_rng(int device, uint64_t seed, size_t n, const uint32_t* keys, uint32_t n_draws, float* out) {
    if (!keys || !out) return rt_fail(RT_ERR_INVALID, "rt_probe_rng: null argument");
    if (n == 0 || n_draws == 0) return RT_OK;
    int rc = select_device(device);
    if (rc != RT_OK) return rc;
    DevBuf k, o;
    UP(k, keys, n * 8);
    HIP_TRY(o.alloc(n * n_draws * 4));
    probe_rng_kernel<<<PROBE_GRID(n)>>>(seed, n, k.as<uint32_t>(), n_draws, o.as<float>());
    FINISH();
    DOWN(out, o, n * n_draws * 4);
    return RT_OK;
}


__global__ void probe_math_kernel(int fn, size_t n, const float* a, const float* b, float* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = fn == 0 ? rt_logf(a[i]) : fn == 1 ? rt_sinf(a[i]) : fn == 2 ? rt_acosf(a[i]) : rt_atan2f(a[i], b[i]);
}
extern "C" int rt_probe_math(int device, int fn, size_t n, const float* a, const float* b, float* out) {
    if (!a || !b || !out) return rt_fail(RT_ERR_INVALID, "rt_probe_math: null argument");
    if (fn < 0 || fn > 3) return rt_fail(RT_ERR_INVALID, "rt_probe_math: unknown function %d", fn);
    if (n == 0) return RT_OK;
    int rc = select_device(device);
    if (rc != RT_OK) return rc;
    DevBuf da, db, o;
    UP(da, a, n * 4); UP(db, b, n * 4);
    HIP_TRY(o.alloc(n * 4));
    probe_math_kernel<<<PROBE_GRID(n)>>>(fn, n, da.as<float>(), db.as<float>(), o.as<float>());
    FINISH();
    DOWN(out, o, n * 4);
    return RT_OK;
}

// The device half of the math vocabulary (csrc/rt_math.hpp) over arrays: the functions the fixtures tests/golden/glm_*.f32 —
// generated by the REFERENCE's vendored GLM + glm_utils.h (oracle/ref_glm_probe.cpp) — cover, plus Ray::at / isBackfacing
// (tests/golden/ref_ray_*, from the reference's ray_data.cuh).  The one direct reference -> HIP check there is.
__global__ void probe_glm_kernel(int fn, size_t n, uint32_t nin, uint32_t nout, const float* __restrict__ in, float* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* a = in + i * nin;
    float* o = out + i * nout;
    switch (fn) {
        case 0: o[0] = dot(ld3(a), ld3(a + 3)); break;
        case 1: st3(o, cross(ld3(a), ld3(a + 3))); break;
        case 2: st3(o, normalize(ld3(a))); break;
        case 3: st3(o, reflect(ld3(a), ld3(a + 3))); break;
        case 4: st3(o, refract(ld3(a), ld3(a + 3), a[6])); break;
        case 5: st3(o, mix(ld3(a), ld3(a + 3), a[6])); break;
        case 6: o[0] = mix(a[0], a[1], a[2]); break;
        case 7: st3(o, glm_min(ld3(a), ld3(a + 3))); break;
        case 8: st3(o, glm_max(ld3(a), ld3(a + 3))); break;
        case 9: o[0] = comp_max(ld3(a)); break;
        case 10: o[0] = comp_min(ld3(a)); break;
        case 11: st3(o, clamp01_sqrt(ld3(a))); break;
        case 12: o[0] = near_zero(ld3(a)) ? 1.0f : 0.0f; break;
        case 13: o[0] = length2(ld3(a)); break;
        case 14: st3(o, linear_interpolate(ld3(a), ld3(a + 3), a[6])); break;
        case 15: o[0] = radians(a[0]); break;
        default: {  // 16: Ray::at (ray_data.cuh:14) + isBackfacing (ray_data.cuh:44-46): (o, d, t, normal) -> (at, backfacing)
            Ray r; r.o = ld3(a); r.d = ld3(a + 3); r.time = 0.0f;
            st3(o, ray_at(r, a[6]));
            o[3] = dot(r.d, ld3(a + 7)) > 0 ? 1.0f : 0.0f;
        }
    }
}
extern "C" int rt_probe_glm(int device, int fn, size_t n, const float* in, float* out) {
    static const uint32_t shape[17][2] = {{6, 1}, {6, 3}, {3, 3}, {6, 3}, {7, 3}, {7, 3}, {3, 1}, {6, 3}, {6, 3}, {3, 1}, {3, 1}, {3, 3}, {3, 1},
                                          {3, 1}, {7, 3}, {1, 1}, {10, 4}};
    if (!in || !out) return rt_fail(RT_ERR_INVALID, "rt_probe_glm: null argument");
    if (fn < 0 || fn > 16) return rt_fail(RT_ERR_INVALID, "rt_probe_glm: unknown function %d", fn);
    if (n == 0) return RT_OK;
    int rc = select_device(device);
    if (rc != RT_OK) return rc;
    const uint32_t nin = shape[fn][0], nout = shape[fn][1];
    DevBuf di, dout;
    UP(di, in, n * nin * 4);
    HIP_TRY(dout.alloc(n * nout * 4));
    probe_glm_kernel<<<PROBE_GRID(n)>>>(fn, n, nin, nout, di.as<float>(), dout.as<float>());
    FINISH();
    DOWN(out, dout, n * nout * 4);
    return RT_OK;
}

// ---------------------------------------------------------------------------------------------
// verification of rt_fastdiv.hpp
// ---------------------------------------------------------------------------------------------
__global__ void probe_aabb_regular_kernel(size_t n, const float* boxes, const float* rays, const float* maxd, int32_t* regular,
                                          int32_t* hit, float* dist) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ray r;
    r.o = ld3(rays + 6 * i); r.d = ld3(rays + 6 * i + 3); r.time = 0.0f;
    f3 bmin = ld3(boxes + 6 * i), bmax = ld3(boxes + 6 * i + 3);
    bool reg = ray_is_regular(r) && coord_is_regular(bmin.x) && coord_is_regular(bmin.y) && coord_is_regular(bmin.z) &&
               coord_is_regular(bmax.x) && coord_is_regular(bmax.y) && coord_is_regular(bmax.z);
    regular[i] = reg ? 1 : 0;
    float d = 0.0f;
    bool h = false;
    if (reg) {
        f3 inv_d = mk3(rcp_exact_regular(r.d.x), rcp_exact_regular(r.d.y), rcp_exact_regular(r.d.z));  // as the render kernel does
        h = aabb_intersects_regular(bmin, bmax, r, inv_d, maxd[i], d);
    }
    hit[i] = h ? 1 : 0;
    dist[i] = d;
}

extern "C" int rt_probe_aabb_regular(int device, size_t n, const float* boxes, const float* rays, const float* max_dist,
                                     int32_t* out_regular, int32_t* out_hit, float* out_dist) {
    if (!boxes || !rays || !max_dist || !out_regular || !out_hit || !out_dist) return rt_fail(RT_ERR_INVALID, "rt_probe_aabb_regular: null argument");
    if (n == 0) return RT_OK;
    int rc = select_device(device);
    if (rc != RT_OK) return rc;
    DevBuf b, r, m, g, h, d;
    UP(b, boxes, n * 24); UP(r, rays, n * 24); UP(m, max_dist, n * 4);
    HIP_TRY(g.alloc(n * 4)); HIP_TRY(h.alloc(n * 4)); HIP_TRY(d.alloc(n * 4));
    probe_aabb_regular_kernel<<<PROBE_GRID(n)>>>(n, b.as<float>(), r.as<float>(), m.as<float>(), g.as<int32_t>(), h.as<int32_t>(), d.as<float>());
    FINISH();
    DOWN(out_regular, g, n * 4); DOWN(out_hit, h, n * 4); DOWN(out_dist, d, n * 4);
    return RT_OK;
}

// one block per divisor significand; its 256 threads sweep all 2^23 numerator significands
__global__ __launch_bounds__(256) void selftest_fastrcp_kernel(unsigned long long* counts, uint32_t* example) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long n = 0, bad = 0;
    for (uint64_t u = i; u < (1ull << 32); u += stride) {
        const uint32_t bits = (uint32_t)u;
        const uint32_t e = (bits >> 23) & 0xffu;
        if (e < 127u - 40u || e > 127u + 39u) continue;
        const float x = __uint_as_float(bits);
        n++;
        if (__float_as_uint(rcp_exact_regular(x)) != __float_as_uint(1.0f / x)) { bad++; *example = bits; }
    }
    atomicAdd(counts + 0, n);
    atomicAdd(counts + 1, bad);
}

extern "C" int rt_selftest_fastrcp(int device, uint64_t* checked, uint64_t* mismatches, uint32_t* example) {
    if (!checked || !mismatches || !example) return rt_fail(RT_ERR_INVALID, "rt_selftest_fastrcp: null argument");
    int rc = select_device(device);
    if (rc != RT_OK) return rc;
    DevBuf counts, ex;
    HIP_TRY(counts.alloc(16));
    HIP_TRY(ex.alloc(4));
    HIP_TRY(hipMemset(counts.p, 0, 16));
    HIP_TRY(hipMemset(ex.p, 0, 4));
    selftest_fastrcp_kernel<<<8192, 256>>>(counts.as<unsigned long long>(), ex.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long h[2];
    HIP_TRY(hipMemcpy(h, counts.p, 16, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(example, ex.p, 4, hipMemcpyDeviceToHost));
    *checked = h[0];
    *mismatches = h[1];
    return RT_OK;
}

__global__ __launch_bounds__(256) void selftest_fastdiv_kernel(uint32_t first_den, uint32_t num_exp_bits, uint32_t den_exp_bits,
                                                               unsigned long long* mismatches, uint32_t* example) {
    uint32_t md = first_den + blockIdx.x;
    float d = __uint_as_float(den_exp_bits | md);
    float r = 1.0f / d;
    float rl = rcp_low_word(d, r);
    uint32_t bad = 0;
    uint32_t bad_n = 0;
    for (uint32_t mn = threadIdx.x; mn < (1u << 23); mn += 256u) {
        float n = __uint_as_float((num_exp_bits & ~1u) | mn);
        float q = (num_exp_bits & 1u) ? fast_div_exact4(n, d, r, rl) : fast_div_exact(n, d, r);   // bit 0 of the exponent word = mode
        float ref = n / d;
        if (__float_as_uint(q) != __float_as_uint(ref)) { bad++; bad_n = __float_as_uint(n); }
    }
    if (bad) {
        atomicAdd(mismatches, (unsigned long long)bad);
        example[0] = bad_n;
        example[1] = __float_as_uint(d);
    }
}

extern "C" int rt_selftest_fastdiv(int device, uint32_t first_den, uint32_t n_den, int32_t num_exp, int32_t den_exp,
                                   uint64_t* mismatches, uint32_t example[2]) {
    if (!mismatches || !example) return rt_fail(RT_ERR_INVALID, "rt_selftest_fastdiv: null argument");
    if (first_den >= (1u << 23) || n_den == 0 || n_den > (1u << 23) - first_den) return rt_fail(RT_ERR_INVALID, "rt_selftest_fastdiv: significand range out of bounds");
    if (num_exp < -126 || num_exp > 127 || den_exp < -126 || den_exp > 127) return rt_fail(RT_ERR_INVALID, "rt_selftest_fastdiv: exponent out of range");
    int rc = select_device(device);
    if (rc != RT_OK) return rc;
    DevBuf cnt, ex;
    HIP_TRY(cnt.alloc(8)); HIP_TRY(ex.alloc(8));
    HIP_TRY(hipMemset(cnt.p, 0, 8)); HIP_TRY(hipMemset(ex.p, 0, 8));
    selftest_fastdiv_kernel<<<n_den, 256>>>(first_den, (uint32_t)(num_exp + 127) << 23, (uint32_t)(den_exp + 127) << 23,
                                            cnt.as<unsigned long long>(), ex.as<uint32_t>());
    FINISH();
    DOWN(mismatches, cnt, 8); DOWN(example, ex, 8);
    return RT_OK;
}

extern "C" int rt_selftest_fastdiv4(int device, uint32_t first_den, uint32_t n_den, int32_t num_exp, int32_t den_exp,
                                   uint64_t* mismatches, uint32_t example[2]) {
    if (!mismatches || !example) return rt_fail(RT_ERR_INVALID, "rt_selftest_fastdiv4: null argument");
    if (first_den >= (1u << 23) || n_den == 0 || n_den > (1u << 23) - first_den) return rt_fail(RT_ERR_INVALID, "rt_selftest_fastdiv4: significand range out of bounds");
    if (num_exp < -126 || num_exp > 127 || den_exp < -126 || den_exp > 127) return rt_fail(RT_ERR_INVALID, "rt_selftest_fastdiv4: exponent out of range");
    int rc = select_device(device);
    if (rc != RT_OK) return rc;
    DevBuf cnt, ex;
    HIP_TRY(cnt.alloc(8)); HIP_TRY(ex.alloc(8));
    HIP_TRY(hipMemset(cnt.p, 0, 8)); HIP_TRY(hipMemset(ex.p, 0, 8));
    selftest_fastdiv_kernel<<<n_den, 256>>>(first_den, ((uint32_t)(num_exp + 127) << 23) | 1u, (uint32_t)(den_exp + 127) << 23,
                                            cnt.as<unsigned long long>(), ex.as<uint32_t>());
    FINISH();
    DOWN(mismatches, cnt, 8); DOWN(example, ex, 8);
    return RT_OK;
}


// box_pair_filtered vs the exact decisions, on regular inputs only
__global__ void probe_boxpair_kernel(size_t n, const float* boxes, const float* rays, const float* maxd, int32_t* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ray r;
    r.o = ld3(rays + 6 * i); r.d = ld3(rays + 6 * i + 3); r.time = 0.0f;
    const float* b = boxes + 12 * i;
    bool reg = ray_is_regular(r);
    for (int k = 0; k < 12; k++) reg = reg && coord_is_regular(b[k]);
    for (int k = 0; k < 3; k++) reg = reg && b[k] <= b[3 + k] && b[6 + k] <= b[9 + k];
    int32_t* o = out + 8 * i;
    for (int k = 0; k < 8; k++) o[k] = 0;
    o[0] = reg ? 1 : 0;
    if (!reg) return;
    f3 inv_d = mk3(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
    BoxPairDecision d = box_pair_filtered(ld3(b), ld3(b + 3), ld3(b + 6), ld3(b + 9), r, inv_d, maxd[i]);
    float dl = RT_MISS_DIST, dr = RT_MISS_DIST;
    bool hl = aabb_intersects(ld3(b), ld3(b + 3), r, maxd[i], dl);
    bool hr = aabb_intersects(ld3(b + 6), ld3(b + 9), r, maxd[i], dr);
    o[1] = d.uncertain; o[2] = d.hit_left; o[3] = d.hit_right; o[4] = d.swap;
    o[5] = hl; o[6] = hr; o[7] = dl > dr;
}

extern "C" int rt_probe_boxpair_filtered(int device, size_t n, const float* boxes, const float* rays, const float* max_dist, int32_t* out) {
    if (!boxes || !rays || !max_dist || !out) return rt_fail(RT_ERR_INVALID, "rt_probe_boxpair_filtered: null argument");
    if (n == 0) return RT_OK;
    int rc = select_device(device);
    if (rc != RT_OK) return rc;
    DevBuf b, r, m, o;
    UP(b, boxes, n * 48); UP(r, rays, n * 24); UP(m, max_dist, n * 4);
    HIP_TRY(o.alloc(n * 32));
    probe_boxpair_kernel<<<PROBE_GRID(n)>>>(n, b.as<float>(), r.as<float>(), m.as<float>(), o.as<int32_t>());
    FINISH();
    DOWN(out, o, n * 32);
    return RT_OK;
}

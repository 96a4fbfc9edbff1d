// rt_stream_kernel.hpp — the production render path: a persistent, sample-streaming wavefront kernel
// with the whole BVH resident in LDS, plus the in-order resolve kernel.
//
// Reference shape (main/src/Renderer.cu:183-217): one thread per PIXEL, serial spp loop, 48 B of RNG
// state per pixel in global memory, two virtual calls per bounce, BVH nodes fetched from global memory
// by pointer.  MI355X shape:
//
//  * one work-item owns one pixel-SAMPLE.  The rank's samples of a pass form one linear index space
//    n = (block * pass_spp + s) * 64 + pixel-in-block (a block = the 64 pixels of an 8x8 tile), so the
//    64 lanes of a wavefront mostly hold one sample each of the 64 pixels of a tile (coherent primary
//    rays) and n is also the slot of the sample in the HBM sample buffer.  Waves are persistent: each pulls chunks of
//    `chunk` (<= 1024) consecutive sample indices from one global counter and hands them to its lanes with a
//    ballot + prefix popcount the moment a lane's path ends (sample regeneration), so path-length
//    divergence (1..50 bounces) does not idle lanes.
//  * the lane program is the reference's: primary ray, then trace / scatter per bounce with
//    BVH::ClosestIntersection's exact order of box tests, culling and pushes.  What is re-scheduled is
//    only WHICH lanes run WHICH phase together: the wave runs "inner node" steps while most lanes sit at
//    an inner node, services lanes that reached a leaf or finished a trace when enough have piled up
//    (wave ballots), and never lets one lane's long traversal hold 63 finished ones.
//  * the scene is staged once per workgroup into LDS (coalesced 16-B loads of one packed blob): 76-B
//    "wide" inner nodes that carry BOTH child boxes as (min, max, min) triples and both child references (so a visit
//    is one address and leaves need no node fetch at all; the triples turn the slab test's per-axis min/max into a
//    per-ray address offset), 16-B sphere records and 16-B (centre1, material) records.  The per-lane traversal stack
//    lives in LDS too, sized from the tree's depth.  Worlds that do not fit keep their records in global memory / L2
//    (BIG instantiation, 32-bit references).
//  * gfx950 issues fp32 add/mul/fma and simple integer ops in 2 cycles per wave, compares / selects / min / max in 4,
//    and scalar instructions are NOT hidden behind the vector issue (tools/bench_valu_issue.hip): the hot loop is
//    straight-line code built from the cheap class, one compare per phase test, exact division without dividing.
//  * the counter-based RNG keeps no state in memory; every sample writes its radiance (12 B) to an HBM
//    sample buffer laid out [pixel-block][sample][64 pixels], and `resolve_kernel` adds them IN SAMPLE
//    ORDER per pixel — the same order as the reference's `radiance +=` loop — so the framebuffer is
//    bit-identical to the CPU oracle's, and independent of scheduling and of the number of GPUs.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#include "rt06.h"
#include "rt_device_funcs.hpp"
#include "rt_fastdiv.hpp"
#include "rt_internal.hpp"
#include "rt_layout.hpp"
#include "rt_render_kernels.hpp"

#define RT_WORLD_BVH_QUEUE 3     // internal WORLD mode of render_kernel_stream: an RT_WORLD_BVH world walked by each lane on its own — the distance-sorted queue (RT_TRAVERSAL_QUEUE) or the 4-wide walk (RT_TRAVERSAL_WIDE4)
#define RT_CHUNK_MAX 768u        // sample indices a wave pulls per atomic (the host shrinks it for small frames / shards); 768 measured 0.35 ms ahead of 1024 and of 512 on config 2
// scheduling thresholds (lanes of 64); overridable per renderer for tuning (RT06_TUNE=keep,shade,leaf)
#define RT_INNER_KEEP 36         // keep iterating inner-node steps while at least this many lanes want one (round 4, with the cheaper hot step: 36 / 52 / 4 is 0.3-1.2 % ahead of 40 / 56 / 4 on all four workloads)
#define RT_SHADE_MIN 52          // run the shade/regenerate phase once this many lanes wait for it
#define RT_LEAF_MIN 4            // run the leaf phase once this many lanes sit at a leaf (or nobody is at an inner node)

// A primary-ray record is read exactly once: a non-temporal load, so that this 23 GB stream does not push the partially written lines of
// the sample buffer out of the L2 before their neighbours arrive.  Measured (config 2, round 3): HBM write traffic of the kernel
// 9.79 GB -> 6.63 GB for 5.76 GB of samples (1.70x -> 1.15x), same time (EXPERIMENTS.md).
typedef float rt_nt_f4 __attribute__((ext_vector_type(4)));
typedef uint32_t rt_nt_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 rt_load_once(const float4* p) {
    const rt_nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const rt_nt_f4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint4 rt_load_once(const uint4* p) {
    const rt_nt_u4 v = __builtin_nontemporal_load(reinterpret_cast<const rt_nt_u4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
#define RT_LOAD_ONCE(ptr) rt_load_once(ptr)
// The generator's stores and the resolve kernel's loads are streams too (written once / read once): non-temporal, -0.13 ms per config-2 frame
// (primary 4.03 -> 3.94 ms, resolve 0.99 -> 0.95 ms).
typedef float rt_nts_f4 __attribute__((ext_vector_type(4)));
typedef uint32_t rt_nts_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void rt_store_stream(float4* p, float4 v) { rt_nts_f4 w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, reinterpret_cast<rt_nts_f4*>(p)); }
__device__ __forceinline__ void rt_store_stream(uint4* p, uint4 v) { rt_nts_u4 w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, reinterpret_cast<rt_nts_u4*>(p)); }
__device__ __forceinline__ float rt_load_stream(const float* p) { return __builtin_nontemporal_load(p); }

#define RT_SAMPLE_BYTES 12u
__device__ __forceinline__ void store_sample(float4* samples, uint32_t n, float x, float y, float z) {
    float* o = reinterpret_cast<float*>(samples) + (size_t)n * 3u;
    o[0] = x; o[1] = y; o[2] = z;
}
__device__ __forceinline__ f3 load_sample(const float4* samples, size_t n) {
    const float* q = reinterpret_cast<const float*>(samples) + n * 3u;
    return mk3(rt_load_stream(q), rt_load_stream(q + 1), rt_load_stream(q + 2));
}

struct StreamParams {
    uint32_t width, height, spp, max_depth;
    uint64_t seed;
    rt_camera cam;
    TileMap tm;
    PackedSceneRef scene;
    uint32_t pass_first_s;   // first sample index of this pass
    uint32_t pass_spp;       // samples per pixel in this pass
    uint32_t total;          // n_local_pixels * pass_spp
    uint32_t inner_keep, shade_min, leaf_min;
    uint32_t chunk;          // sample indices per work-queue fetch
    float4* samples;         // [n_local_pixels/64][pass_spp][64] radiance, RT_SAMPLE_BYTES = 12 bytes per sample index n (one global_store_dwordx3;
                             // 16-byte slots were measured slower and MORE HBM write traffic: EXPERIMENTS.md)
    // primary rays of the pass, written by primary_rays_kernel and consumed by sample regeneration, indexed by n:
    float4* prim_o;          // (ray origin, ray time)
    float4* prim_d;          // (ray direction, -)
    uint4* prim_rng;         // the sample's RNG state after the camera's draws; all zero (never a valid state) marks a padding pixel
    uint32_t* work_counter;
    DeviceWorld world;       // WORLD == RT_WORLD_BVH_QUEUE only: the flat world as BVH::ClosestIntersection's queue walk reads it (BVH.cu:17-49, :80-86)
#ifdef RT_PHASE_TIMERS
    unsigned long long* phase_acc;  // development build only: [0..15] cycles per phase, [16..31] visits (summed over waves)
#endif
};

__device__ __forceinline__ uint32_t lane_rank(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// pixel origin of the 8x8 tile that local 64-pixel block `blk` covers; false for a padding block
__device__ __forceinline__ bool block_origin(const TileMap& tm, uint32_t blk, uint32_t& x0, uint32_t& y0) {
    uint32_t gt = blk * tm.world_size + tm.rank;
    uint32_t ty = gt / tm.tiles_x;
    x0 = (gt - ty * tm.tiles_x) * RT_TILE;
    y0 = ty * RT_TILE;
    return gt < tm.n_tiles;
}

// A lane's status lives in `cur` alone (compares and selects are half-rate on gfx950; one register, one compare):
//   cur <  RT_REF_IRR              at an inner node, ray in the fast-division class            } tracing
//   cur <  RT_REF_LEAF             at an inner node (marked RT_REF_IRR: ray outside the class) }  (cur < K_SHADE)
//   cur <  K_SHADE                 at a leaf: RT_REF_LEAF | code (code <= 0x7ffe)              }
//   cur == K_SHADE                 trace finished, waiting for the shade phase
//   cur == K_NEED                  path finished, waiting for a new sample
//   cur == K_OFF                   no samples left
//   cur == K_START                 got a new ray in this round; its trace begins at the end of the round
// Values (K_* constants inside the kernel): 16-bit references: SHADE 0xffff (fits a stack entry: the sentinel at the bottom of
// the stack; leaf codes stop at 0x7ffe), NEED 0x20000, OFF 0x30000, START 0x40000; 32-bit references (BIG): the top four
// values 0xfffffffc .. 0xffffffff.  "Tracing" is `cur < K_SHADE` either way.

// The (near, far) plane pairs of both child boxes of wide node `idx` and its two child references.  kx/ky/kz: 0 where the
// ray direction component is >= 0 (near = box min) or the ray is outside the fast class, 4 where it is negative.
struct WideNodeData {
    float lnx, lny, lnz, lfx, lfy, lfz, rnx, rny, rnz, rfx, rfy, rfz;
    uint32_t left, right;
};
// BIG: nodes below n_top come from their LDS copy (`top`), the others from global memory — the global path is bound by
// the L1's tag lookups, which only the lanes that really go to memory consume.
template <bool BIG>
__device__ __forceinline__ WideNodeData fetch_wide_node(const char* nodes, const uint4* top, uint32_t n_top, uint32_t idx,
                                                        uint32_t kx, uint32_t ky, uint32_t kz) {
    WideNodeData n;
    if (BIG) {
        float4 a, b, c;
        uint4 r;
        if (idx < n_top) {
            const float4* q = reinterpret_cast<const float4*>(top + idx * (RT_NODE_DWORDS_BIG / 4u));
            a = q[0]; b = q[1]; c = q[2];
            r = reinterpret_cast<const uint4*>(q)[3];
        } else {
            const float4* q = reinterpret_cast<const float4*>(nodes + idx * (RT_NODE_DWORDS_BIG * 4u));
            a = q[0]; b = q[1]; c = q[2];
            r = reinterpret_cast<const uint4*>(q)[3];
        }
        const bool sx = kx != 0u, sy = ky != 0u, sz = kz != 0u;
        n.lnx = sx ? a.w : a.x; n.lfx = sx ? a.x : a.w;
        n.lny = sy ? b.x : a.y; n.lfy = sy ? a.y : b.x;
        n.lnz = sz ? b.y : a.z; n.lfz = sz ? a.z : b.y;
        n.rnx = sx ? c.y : b.z; n.rfx = sx ? b.z : c.y;
        n.rny = sy ? c.z : b.w; n.rfy = sy ? b.w : c.z;
        n.rnz = sz ? c.w : c.x; n.rfz = sz ? c.x : c.w;
        n.left = r.x; n.right = r.y;
    } else {
        // one 32-bit byte offset per access: (near, far) are consecutive dwords of the (min, max, min) triples
        const uint32_t nb = idx * RT_NODE_BYTES;
        const float* px = reinterpret_cast<const float*>(nodes + (nb + kx));
        const float* py = reinterpret_cast<const float*>(nodes + (nb + ky));
        const float* pz = reinterpret_cast<const float*>(nodes + (nb + kz));
        n.lnx = px[0]; n.lfx = px[1]; n.rnx = px[9]; n.rfx = px[10];
        n.lny = py[3]; n.lfy = py[4]; n.rny = py[12]; n.rfy = py[13];
        n.lnz = pz[6]; n.lfz = pz[7]; n.rnz = pz[15]; n.rfz = pz[16];
        const uint32_t refs = reinterpret_cast<const uint32_t*>(nodes + nb)[RT_NODE_REFS];
        n.left = refs & 0xffffu; n.right = refs >> 16;
    }
    return n;
}

// EXACT = true : box tests use aabb_intersects() verbatim (IEEE division, GLM min/max).
// EXACT = false: rays classified "regular" use the 4-instruction correctly-rounded division (two-word reciprocal) and
//                near/far plane selection by address (rt_fastdiv.hpp) — identical decisions, proven + exhaustively
//                verified; other rays carry marked references and are stepped by a verbatim loop.
// FILTER = true (variant 4, experimental): decide the two box tests of a visit from one-multiply plane
//                parameters with a safety margin (box_pair_filtered) and fall back to exact quotients only
//                for near-ties; sound and bit-identical, but not faster yet because ~1.5 % of visits are ties.
// WORLD: RT_WORLD_BVH (default), RT_WORLD_LIST (HittableList: bounds pre-test, then every sphere in order;
//        a reference is RT_REF_LEAF | primitive index and the "traversal" is the leaf phase alone) or
//        RT_WORLD_NODE_TREE (bvh_node: a node is tested against ITS OWN box when visited, then left, then right), or
//        RT_WORLD_BVH_QUEUE (the reference's disabled distance-sorted queue, BVH.cu:17-49: the frontier is one sorted list per ray, so
//        there is no wave-level hot loop to share — a lane walks its whole trace when the trace begins (bvh_closest_intersection_queue,
//        the function the probes and the baseline kernel run) and joins the others again at the shade phase; what the streaming kernel
//        adds for such a world is the in-order resolve, i.e. a framebuffer bit-identical to the oracle's, and the sample streaming).
// EXT >= 1: the scene uses features the reference does not have (quads, diffuse lights, a constant background, constant
//        media; EXT = 2 adds the two textured materials — Perlin marble and the image texture — whose code is large enough
//        to cost the other kernels registers):
//        leaf codes >= sphere_codes are quads, emitted radiance is accumulated along the path (the reference's
//        commented `accum_radiance`), the miss colour may be a constant.  A separate instantiation, so the
//        reference-feature kernels carry none of it.
// BIG: the records stay in global memory / L2 (the image does not fit the LDS); WIDE: 32-bit references (a BIG world with 2^14
//      inner nodes or 2^15 leaf codes or more) — a BIG world whose references fit 16 bits keeps the narrow encoding, which halves
//      the per-lane stacks and leaves that much more of the LDS for the top of the tree.
// TOL (variant 6, opt-in): the hot loop's plane parameters are (b - o) * RN(1/d) instead of the exact quotients — inside north_star's |delta| < 1e-3,
//      not bit-exact by construction (rt_fastdiv.hpp: slab_near_far_tolerant).  LDS-resident RT_WORLD_BVH worlds of the reference's feature set (EXT == 0) only.
template <bool EXACT, bool FILTER, int BLOCK, int WORLD = RT_WORLD_BVH, int EXT = 0, bool BIG = false, bool WIDE = BIG, bool TOL = false>
__global__ __launch_bounds__(BLOCK, BLOCK / 128) void render_kernel_stream(StreamParams p) {
    extern __shared__ uint4 lds[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;

    // reference encoding and lane status codes (see the header of this file)
    using ref_t = typename std::conditional<WIDE, uint32_t, uint16_t>::type;
    constexpr uint32_t K_LEAF = WIDE ? RT_REF_LEAF_BIG : RT_REF_LEAF, K_IRR = WIDE ? RT_REF_IRR_BIG : RT_REF_IRR;
    constexpr uint32_t K_SHADE = WIDE ? 0xfffffffcu : 0xffffu, K_NEED = WIDE ? 0xfffffffdu : 0x20000u;
    constexpr uint32_t K_OFF = WIDE ? 0xfffffffeu : 0x30000u, K_START = WIDE ? 0xffffffffu : 0x40000u;
    static_assert(BIG || !WIDE, "an LDS-resident image always has 16-bit references");

    // ---- the scene: staged into the LDS with coalesced 16-B loads, or (BIG) left in global memory / L2 -------------
    const uint4* scene_base;
    if (BIG) {
        scene_base = p.scene.blob;
        for (uint32_t i = tid; i < p.scene.n_top * (RT_NODE_DWORDS_BIG / 4u); i += BLOCK) lds[i] = p.scene.blob[i];   // the top of the tree
        __syncthreads();
    } else {
        for (uint32_t i = tid; i < p.scene.blob_vec4; i += BLOCK) lds[i] = p.scene.blob[i];
        __syncthreads();
        scene_base = lds;
    }
    const char* nodes = reinterpret_cast<const char*>(scene_base);
    const float4* spheres = reinterpret_cast<const float4*>(scene_base + p.scene.off_spheres);
    const float4* extra = reinterpret_cast<const float4*>(scene_base + p.scene.off_extra);
    const float4* mats16 = reinterpret_cast<const float4*>(scene_base + p.scene.off_mats);
    const float4* quads = reinterpret_cast<const float4*>(scene_base + p.scene.off_quads);
    // per-lane traversal stack of references (16 bits, BIG: 32).  Entry k of this lane is stack[k * 64].
    ref_t* stack = reinterpret_cast<ref_t*>(lds + (BIG ? p.scene.n_top * (RT_NODE_DWORDS_BIG / 4u) : p.scene.blob_vec4)) + wave * 64u * p.scene.stack_cap + lane;

    const f3 root_min = mk3(p.scene.root_min[0], p.scene.root_min[1], p.scene.root_min[2]);
    const f3 root_max = mk3(p.scene.root_max[0], p.scene.root_max[1], p.scene.root_max[2]);

    // ---- per-lane path state ------------------------------------------------------------------------
    Ray ray;
    ray.o = mk3(0.0f); ray.d = mk3(0.0f); ray.time = 0.0f;
    f3 inv_d = mk3(0.0f);   // RN(1/d) of a regular ray (EXACT == false)
    f3 inv_lo = mk3(0.0f);  // its low word: inv_d + inv_lo ~= 1/d to 47 bits (fast_div_exact4, FAST_BVH only)
    bool regular = false;
    f3 atten = mk3(0.0f);
    f3 accum_rad = mk3(0.0f);  // EXT only: radiance emitted along the path so far
    Rng rng;
    rng.s0 = 1u; rng.s1 = 0u; rng.s2 = 0u; rng.s3 = 0u; rng.draws = 0u;   // every sample brings its own state (primary_rays_kernel)
    float ray_a = 0.0f;     // dot(ray.d, ray.d): the `a` of every sphere test of the trace (SphereHittable.cuh:17), computed once per ray
    float rec_t = RT_MISS_DIST;
    int32_t rec_code = -1;  // leaf code of the closest hit so far, -1 = none
    uint32_t cur = K_NEED;   // node reference being visited, or the lane's status (see RT_CUR_*)
    ref_t* sp = stack + 64;   // next free entry of this lane's stack (entries are 64 apart; entry 0 is the sentinel)
    *stack = (ref_t)K_SHADE;
    uint32_t kx = 0, ky = 0, kz = 0;  // byte offset (0 / 4) of the (near, far) pair inside an axis triple, per ray
    // FAST_BVH: the default kernel (variant 3).  Inner references of rays outside the fast-division class are marked
    // with K_IRR (an LDS-resident tree has < 2^14 inner nodes) so that the hot loop needs no per-lane branch.
    constexpr bool FAST_BVH = !EXACT && !FILTER && WORLD == RT_WORLD_BVH;
    bool irr_pending = false;         // wave-uniform: some lane is traversing with a ray outside the class
    uint32_t depth = 0;
    uint32_t out_idx = 0;   // == the sample index n: the sample buffer is laid out in index order

    // ---- wave-uniform work pool: sample indices [pool_next, pool_end) ----------------------------------
    // The FIRST chunk of a wave is its own (chunk number = the wave's number in the grid; the host starts the shared counter behind
    // them): 6144 waves asking one counter at once take ~70 us (one word serves ~88 returning atomics per us), which a 1/8 shard feels.
    // (readfirstlane: the wave's number is uniform, which the compiler cannot see in threadIdx.x >> 6 — the pool stays in scalar registers)
    uint32_t pool_next = min((blockIdx.x * (BLOCK / 64u) + (uint32_t)__builtin_amdgcn_readfirstlane((int)wave)) * p.chunk, p.total);
    uint32_t pool_end = min(pool_next + p.chunk, p.total);
    bool pool_dry = false;

// BVH.cu:59-60 / HittableList.cuh:22: root (world) box first, against rec.distance (= _MISS_DIST for a fresh
// payload); bvh_node.cuh:20 tests a node's own box when it is visited, so a tree starts at its root unconditionally.
#define RT_BEGIN_TRACE()                                                   \
    do {                                                                   \
        rec_t = RT_MISS_DIST;                                              \
        rec_code = -1;                                                     \
        if (WORLD == RT_WORLD_BVH_QUEUE) {   /* the whole trace, by this lane alone; leaf code as the shade phase wants it */ \
            HitRec qrec_;                                                  \
            qrec_.distance = RT_MISS_DIST; qrec_.normal = mk3(0.0f); qrec_.prim = -1; qrec_.mat = 0; \
            if (p.world.traversal == RT_TRAVERSAL_WIDE4 ? bvh_closest_intersection_wide4(p.world, ray, qrec_, &rng) \
                                                         : bvh_closest_intersection_queue(p.world, ray, qrec_, &rng)) { \
                const uint32_t qp_ = (uint32_t)qrec_.prim;                 \
                rec_t = qrec_.distance;                                    \
                rec_code = qp_ < p.scene.n_prims ? (int32_t)(qp_ * 2u + ((p.world.prims[qp_].mat & RT_PRIM_MOVING) ? 1u : 0u)) \
                                                 : (int32_t)(p.scene.sphere_codes + (qp_ - p.scene.n_prims)); \
            }                                                              \
            cur = K_SHADE;                                                 \
            break;                                                         \
        }                                                                  \
        ray_a = dot(ray.d, ray.d);                                         \
        if (!EXACT) {                                                      \
            regular = ray_is_regular(ray);                                 \
            inv_d = mk3(rcp_exact_regular(ray.d.x), rcp_exact_regular(ray.d.y), rcp_exact_regular(ray.d.z)); /* used by regular rays only */ \
            if (FAST_BVH && !TOL) inv_lo = mk3(rcp_low_word(ray.d.x, inv_d.x), rcp_low_word(ray.d.y, inv_d.y), rcp_low_word(ray.d.z, inv_d.z)); \
            const uint32_t km_ = (regular && FAST_BVH) ? 4u : 0u;          \
            kx = (__float_as_uint(ray.d.x) >> 29) & km_;                   \
            ky = (__float_as_uint(ray.d.y) >> 29) & km_;                   \
            kz = (__float_as_uint(ray.d.z) >> 29) & km_;                   \
        }                                                                  \
        float d_root_;                                                     \
        bool hit_root_;                                                    \
        if (WORLD == RT_WORLD_NODE_TREE) hit_root_ = true;                 \
        else if (EXACT || !regular) hit_root_ = aabb_intersects(root_min, root_max, ray, rec_t, d_root_);         \
        else hit_root_ = root_box_hit_certified(root_min, root_max, ray, inv_d);   /* rec_t is _MISS_DIST here */  \
        if (hit_root_) {                                                   \
            cur = p.scene.root_ref;                                        \
            if (FAST_BVH && !regular && cur < K_LEAF) cur |= K_IRR; \
            sp = stack + 64;                                               \
        } else {                                                           \
            cur = K_SHADE;                                            \
        }                                                                  \
    } while (0)
/* entry 0 of every lane's stack holds K_SHADE: popping an empty stack IS "trace finished", no test needed */ \
#define RT_POP()                      \
    do {                              \
        sp -= 64;                     \
        cur = *sp;                    \
    } while (0)
#define RT_EMIT_DARK() RT_EMIT(EXT ? accum_rad.x : 0.0f, EXT ? accum_rad.y : 0.0f, EXT ? accum_rad.z : 0.0f)
#define RT_EMIT(rx, ry, rz)                                   \
    do {                                                      \
        store_sample(p.samples, out_idx, (rx), (ry), (rz));     \
        cur = K_NEED;                                    \
    } while (0)

#ifdef RT_PHASE_TIMERS
    unsigned long long pt_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pt_last_ = __builtin_readcyclecounter();
    uint32_t pc_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#ifdef RT_TRACE_HIST
    uint32_t tr_steps_ = 0, tr_leafs_ = 0;   // per lane: inner-node steps / leaf tests of the current trace (histogram for tools/sched_model.py)
#endif
#define RT_PT(i) do { unsigned long long n_ = __builtin_readcyclecounter(); pt_[i] += n_ - pt_last_; pt_last_ = n_; pc_[i]++; } while (0)
#else
#define RT_PT(i)
#endif
    for (;;) {
        RT_PT(7);
        // ================= phase 1: inner-node steps (BVH.cu:76-97) ==================================
        if (FAST_BVH) {
            // Hot loop of the default kernel: lanes whose ray is in the fast-division class (all but a handful).  The
            // step is straight-line code; lanes outside the class carry K_IRR in their inner references, so they
            // fail `cur < K_IRR` here and are stepped by the verbatim loop below.
            bool at_inner = cur < K_IRR;
            if (__ballot(at_inner) != 0ull) {
                uint32_t n_inner_lanes;
                // inner_keep >= 1 (host); once the queue is dry the wave only drains its last paths: no reason to leave early.  The threshold is
                // wave-uniform; readfirstlane tells the compiler so (a scalar compare and branch instead of a vector compare and an exec-mask loop exit)
                const uint32_t keep_now = (uint32_t)__builtin_amdgcn_readfirstlane((int)(pool_dry ? 1u : p.inner_keep));
                // Certified far planes (rt_fastdiv.hpp) pay where boxes have thickness.  The box of an axis-aligned quad is 1e-4 thick: the near and far
                // parameters of a ray that hits it agree to ~2e-7, inside the certificate's margin, so nearly every hit box would take the exact redo
                // (measured: Cornell box -6 %, Book-2 final scene -5 %).  Worlds with quads therefore keep the twelve exact quotients; the reference's
                // feature set has no quads (EXT == 0: decided at compile time), the EXT kernels decide per world (wave-uniform).
                const bool far_certified = EXT == 0 || p.scene.n_quads == 0u;
                do {
                    if (at_inner) {
                        const WideNodeData nd = fetch_wide_node<BIG>(nodes, lds, p.scene.n_top, cur, kx, ky, kz);
                        const uint32_t left_idx = nd.left, right_idx = nd.right;
                        float tl, tr;
                        bool hl, hr;
                        if (TOL) {
                            hl = slab_near_far_tolerant(nd.lnx, nd.lny, nd.lnz, nd.lfx, nd.lfy, nd.lfz, ray, inv_d, rec_t, tl);
                            hr = slab_near_far_tolerant(nd.rnx, nd.rny, nd.rnz, nd.rfx, nd.rfy, nd.rfz, ray, inv_d, rec_t, tr);
                        } else if (!far_certified) {   // a world with quads: all twelve parameters as exact quotients
                            hl = slab_near_far_regular(nd.lnx, nd.lny, nd.lnz, nd.lfx, nd.lfy, nd.lfz, ray, inv_d, inv_lo, rec_t, tl);
                            hr = slab_near_far_regular(nd.rnx, nd.rny, nd.rnz, nd.rfx, nd.rfy, nd.rfz, ray, inv_d, inv_lo, rec_t, tr);
                        } else {
                            // near parameters: exact quotients (tl, tr are compared with each other and with rec.distance); far parameters: products,
                            // with the `tmin <= tmax` decisions certified (rt_fastdiv.hpp: CERTIFIED FAR PLANES) — or, rarely and for the whole wave,
                            // redone exactly from a second read of the node
                            tl = slab_near_exact(nd.lnx, nd.lny, nd.lnz, ray, inv_d, inv_lo);
                            tr = slab_near_exact(nd.rnx, nd.rny, nd.rnz, ray, inv_d, inv_lo);
                            float far_l = slab_far_product(nd.lfx, nd.lfy, nd.lfz, ray, inv_d);
                            float far_r = slab_far_product(nd.rfx, nd.rfy, nd.rfz, ray, inv_d);
                            if (__ballot(far_pair_uncertain(tl, far_l, tr, far_r)) != 0ull) {
                                asm volatile("" ::: "memory");   // (a second read of the node: the common path need not keep six plane offsets alive)
                                const WideNodeData n2 = fetch_wide_node<BIG>(nodes, lds, p.scene.n_top, cur, kx, ky, kz);
                                far_l = slab_far_exact(n2.lfx, n2.lfy, n2.lfz, ray, inv_d, inv_lo);
                                far_r = slab_far_exact(n2.rfx, n2.rfy, n2.rfz, ray, inv_d, inv_lo);
                            }
                            hl = tl <= far_l && tl < rec_t && far_l > 0;
                            hr = tr <= far_r && tr < rec_t && far_r > 0;
                        }
                        // BVH.cu:87-96, see the generic loop below: with both boxes hit the far child is pushed and the near
                        // one continues; with one hit it continues; with none the stack is popped.  `left_dist > right_dist`
                        // (missed box = _MISS_DIST) is "right hit and (left missed or tl > tr)".
                        const bool go_right = hr && (!hl || tl > tr);
                        if (hl && hr) {
                            *sp = (ref_t)(go_right ? left_idx : right_idx);
                            sp += 64;
                        }
                        cur = go_right ? right_idx : left_idx;
                        if (!(hl || hr)) RT_POP();
#ifdef RT_TRACE_HIST
                        tr_steps_++;
#endif
                    }
                    at_inner = cur < K_IRR;
                    n_inner_lanes = (uint32_t)__popcll(__ballot(at_inner));
#ifdef RT_PHASE_TIMERS
                    pc_[6]++;
#endif
                } while (n_inner_lanes >= keep_now);
            }
            RT_PT(0);
            if (irr_pending) {   // wave-uniform, rare: rays with a zero / tiny / huge direction or origin component
                for (;;) {
                    const bool at_irr = (cur & (K_LEAF | K_IRR)) == K_IRR;
                    if (__ballot(at_irr) == 0ull) break;
                    if (at_irr) {
                        const WideNodeData nd = fetch_wide_node<BIG>(nodes, lds, p.scene.n_top, cur & (K_IRR - 1u), 0u, 0u, 0u);   // near = min, far = max
                        uint32_t left_idx = nd.left, right_idx = nd.right;
                        if (left_idx < K_LEAF) left_idx |= K_IRR;     // inner references stay marked all the way down
                        if (right_idx < K_LEAF) right_idx |= K_IRR;
                        float left_dist = RT_MISS_DIST, right_dist = RT_MISS_DIST;
                        const bool hl = aabb_intersects(mk3(nd.lnx, nd.lny, nd.lnz), mk3(nd.lfx, nd.lfy, nd.lfz), ray, rec_t, left_dist);
                        const bool hr = aabb_intersects(mk3(nd.rnx, nd.rny, nd.rnz), mk3(nd.rfx, nd.rfy, nd.rfz), ray, rec_t, right_dist);
                        const bool swap_lr = left_dist > right_dist;
                        if (hl && hr) {
                            *sp = (ref_t)(swap_lr ? left_idx : right_idx);
                            sp += 64;
                        }
                        cur = (swap_lr || !hl) ? right_idx : left_idx;
                        if (!(hl || hr)) RT_POP();
                    }
                }
                irr_pending = __ballot(!regular && (cur < K_SHADE)) != 0ull;
            }
            RT_PT(1);
        } else if (WORLD != RT_WORLD_BVH_QUEUE) {   // (a queue world's lanes are never "at a node": their trace ran when it began)
            bool at_inner = cur < K_LEAF;
            uint64_t m_inner = __ballot(at_inner);
            while (m_inner != 0ull) {
                if (at_inner) {
                    // kx/ky/kz are 0 in these kernels: near == box min, far == box max
                    const WideNodeData nd = fetch_wide_node<BIG>(nodes, lds, p.scene.n_top, cur, 0u, 0u, 0u);
                    const float lnx = nd.lnx, lny = nd.lny, lnz = nd.lnz, lfx = nd.lfx, lfy = nd.lfy, lfz = nd.lfz;
                    const float rnx = nd.rnx, rny = nd.rny, rnz = nd.rnz, rfx = nd.rfx, rfy = nd.rfy, rfz = nd.rfz;
                    const uint32_t left_idx = nd.left, right_idx = nd.right;
                    if (WORLD == RT_WORLD_NODE_TREE) {
                        // bvh_node::ClosestIntersection (bvh_node.cuh:19-24): own box, then left subtree, then right
                        float d_own;
                        if (aabb_intersects(mk3(lnx, lny, lnz), mk3(lfx, lfy, lfz), ray, rec_t, d_own)) {
                            *sp = (ref_t)right_idx;
                            sp += 64;
                            cur = left_idx;
                        } else {
                            RT_POP();
                        }
                    } else {
                        const f3 lmin = mk3(lnx, lny, lnz), lmax = mk3(lfx, lfy, lfz), rmin = mk3(rnx, rny, rnz), rmax = mk3(rfx, rfy, rfz);
                        bool hl, hr, swap_lr;
                        if (EXACT || !regular) {
                            float left_dist = RT_MISS_DIST, right_dist = RT_MISS_DIST;
                            hl = aabb_intersects(lmin, lmax, ray, rec_t, left_dist);
                            hr = aabb_intersects(rmin, rmax, ray, rec_t, right_dist);
                            swap_lr = left_dist > right_dist;
                        } else {
                            BoxPairDecision dec = box_pair_filtered(lmin, lmax, rmin, rmax, ray, inv_d, rec_t);
                            hl = dec.hit_left; hr = dec.hit_right; swap_lr = dec.swap;
                            if (dec.uncertain) {
                                // a comparison too close to call (~1e-6 of visits): redo the visit with exact quotients.
                                // The node is re-read (volatile) so that the common path need not keep 12 box
                                // coordinates alive across the filter.
                                const volatile float* vn = reinterpret_cast<const volatile float*>(nodes + cur * RT_NODE_BYTES);   // FILTER kernels are LDS-resident
                                float left_dist = RT_MISS_DIST, right_dist = RT_MISS_DIST;
                                hl = aabb_intersects_regular(mk3(vn[0], vn[3], vn[6]), mk3(vn[1], vn[4], vn[7]), ray, inv_d, rec_t, left_dist);
                                hr = aabb_intersects_regular(mk3(vn[9], vn[12], vn[15]), mk3(vn[10], vn[13], vn[16]), ray, inv_d, rec_t, right_dist);
                                swap_lr = left_dist > right_dist;
                            }
                        }
                        // "assert that left is closer for next step" (BVH.cu:90-93), then push far / near iff
                        // dist < rec.distance (BVH.cu:95-96).  A hit box has dist = tmin < rec.distance by aabb.cuh:41 and
                        // a missed one keeps _MISS_DIST, so the push conditions ARE the hit flags: with both hit the far
                        // child is pushed and the near one (which would be popped straight away) stays in `cur`; with one
                        // hit it becomes `cur` (swap_lr is then exactly "the right one"); with none the stack is popped.
                        if (hl && hr) {
                            *sp = (ref_t)(swap_lr ? left_idx : right_idx);
                            sp += 64;
                        }
                        cur = (swap_lr || !hl) ? right_idx : left_idx;
                        if (!(hl || hr)) RT_POP();
                    }
                }
                at_inner = cur < K_LEAF;
                m_inner = __ballot(at_inner);
                if ((uint32_t)__popcll(m_inner) < p.inner_keep) break;
            }
        }

        // ================= phase 2: leaves (BVH.cu:69-73 -> SphereHittable.cu:56-66 / :91-102) ========
        {
            bool at_leaf = WORLD != RT_WORLD_BVH_QUEUE && (cur - K_LEAF) < (K_SHADE - K_LEAF);
            uint64_t m_leaf = __ballot(at_leaf);
            if (m_leaf != 0ull && ((uint32_t)__popcll(m_leaf) >= p.leaf_min || __ballot(cur < K_LEAF) == 0ull)) {
                if (at_leaf) {
#ifdef RT_TRACE_HIST
                    tr_leafs_++;
#endif
                    uint32_t code = cur & (K_LEAF - 1u);   // BVH / tree: prim * 2 + is_moving;  list: unified primitive index
                    const uint32_t first_quad = (WORLD == RT_WORLD_LIST) ? p.scene.n_prims : p.scene.sphere_codes;
                    if (EXT && code >= first_quad) {
                        // quad::hit ("The Next Week"), reference conventions: see quad_closest_intersection()
                        const float4* qd = quads + (code - first_quad) * 4u;
                        float4 a0 = qd[0], a1 = qd[1], a2 = qd[2], a3 = qd[3];
                        // all four parts in ONE round trip: left alone, the compiler sinks the reads of (Q, D) below the tests that use the normal — three
                        // dependent round trips per quad test, which the global-memory form feels
                        asm volatile("" : : "v"(a0.x), "v"(a0.w));
                        HitRec tmp;
                        tmp.distance = rec_t; tmp.normal = mk3(0.0f); tmp.prim = -1; tmp.mat = 0;
                        if (quad_closest_intersection(mk3(a0.x, a0.y, a0.z), a0.w, mk3(a1.x, a1.y, a1.z), mk3(a1.w, a2.x, a2.y),
                                                      mk3(a2.z, a2.w, a3.x), mk3(a3.y, a3.z, a3.w), 0u, 0, ray, tmp)) {
                            rec_t = tmp.distance;
                            rec_code = (int32_t)(code - first_quad + p.scene.sphere_codes);   // the shade phase's code space
                        }
                        if (WORLD == RT_WORLD_LIST) {  // HittableList.cuh:26-30: every object, in order (the quads follow the spheres)
                            if (code + 1u < p.scene.n_prims + p.scene.n_quads) cur = K_LEAF | (code + 1u);
                            else cur = K_SHADE;
                        } else {
                            RT_POP();
                        }
                    } else {
                    uint32_t prim = (WORLD == RT_WORLD_LIST) ? code : code >> 1;
                    float4 sph = spheres[prim];
                    f3 center = mk3(sph.x, sph.y, sph.z);
                    uint32_t sph_matbits = 0u;  // EXT: a sphere with an isotropic material is a constant medium
                    if (WORLD == RT_WORLD_LIST) {
                        float4 ex = extra[prim];
                        const uint32_t moving = (__float_as_uint(ex.w) >> 28) & 1u;
                        code = prim * 2u + moving;
                        if (moving) center = mix(center, mk3(ex.x, ex.y, ex.z), ray.time);
                        sph_matbits = __float_as_uint(ex.w);
                    } else if (EXT || (code & 1u)) {
                        float4 ex = extra[prim];
                        if (code & 1u) center = mix(center, mk3(ex.x, ex.y, ex.z), ray.time);
                        sph_matbits = __float_as_uint(ex.w);
                    }
                    float t;
                    if (EXT && (sph_matbits >> 29) == RT_MAT_ISOTROPIC)
                        t = medium_sphere_intersection(ray, center, sph.w, mats16[sph_matbits & RT_MAT_INDEX_MASK].w, rec_t, rng);
                    else
                        t = sphere_closest_intersection_a(ray, ray_a, center, sph.w);
                    if (!(t >= rec_t)) {  // `if (t >= rec.distance) return false;`
                        rec_t = t;
                        rec_code = (int32_t)code;
                    }
                    if (WORLD == RT_WORLD_LIST) {  // HittableList.cuh:26-30: every object, in order
                        if (prim + 1u < p.scene.n_prims + (EXT ? p.scene.n_quads : 0u)) cur = K_LEAF | (prim + 1u);
                        else cur = K_SHADE;
                    } else {
                        RT_POP();
                    }
                    }
                }
            }
        }

        RT_PT(2);
        // ================= phase 3: shade finished traces, regenerate finished paths ==================
        // lanes that are not tracing wait for this phase (switched-off lanes count as waiting: near the end of the
        // pass that only makes the phase run a little earlier)
        uint64_t m_trav = __ballot((cur < K_SHADE));
        if (64u - (uint32_t)__popcll(m_trav) < p.shade_min && m_trav != 0ull) continue;

        bool start_trace = false;  // lanes that got a new ray this round begin their trace in ONE place below
        RT_PT(8);
        if (cur == K_SHADE) {  // sample_world's loop body after the trace (Renderer.cu:149-176)
#if defined(RT_PHASE_TIMERS) && defined(RT_TRACE_HIST)   // the histogram's atomics slow the kernel 40x: its own build flag
            atomicAdd(p.phase_acc + 32 + min(tr_steps_, 95u) * 16u + min(tr_leafs_, 15u), 1ull);
            tr_steps_ = 0; tr_leafs_ = 0;
#endif
            if (rec_code < 0) {
                f3 sky;
                if (EXT && p.scene.background == 1u) {
                    sky = mk3(p.scene.background_color[0], p.scene.background_color[1], p.scene.background_color[2]);
                } else {
                    float t = normalize(ray.d).y * 0.5f + 0.5f;
                    sky = linear_interpolate(mk3(0.1f, 0.2f, 0.4f), mk3(0.9f, 0.9f, 0.99f), t);
                }
                f3 rad = atten * sky;
                if (EXT) rad = rad + accum_rad;
                RT_EMIT(rad.x, rad.y, rad.z);
            } else if (!EXT && depth + 1u >= p.max_depth) {
                RT_EMIT(0.0f, 0.0f, 0.0f);  // the scatter of the last allowed bounce cannot reach the sky: result is 0
            } else {
                // ---- what every material needs: hit point, outward normal, material record ----
                const f3 hit_p = ray_at(ray, rec_t);
                f3 normal;
                uint32_t mat_bits;
                if (EXT && (uint32_t)rec_code >= p.scene.sphere_codes) {
                    const float4 qs = quads[p.scene.n_quads * 4u + ((uint32_t)rec_code - p.scene.sphere_codes)];   // the quad's shade record
                    normal = mk3(qs.x, qs.y, qs.z);
                    if (dot(ray.d, normal) > 0) normal = -normal;  // the book's set_face_normal: a quad is two-sided
                    mat_bits = __float_as_uint(qs.w);
                } else {
                    uint32_t prim = (uint32_t)rec_code >> 1;
                    float4 sph = spheres[prim];
                    float4 ex = extra[prim];
                    f3 center = mk3(sph.x, sph.y, sph.z);
                    if ((uint32_t)rec_code & 1u) center = mix(center, mk3(ex.x, ex.y, ex.z), ray.time);
                    normal = (hit_p - center) / sph.w;  // SphereHittable.cu:64 / :100
                    mat_bits = __float_as_uint(ex.w);
                }
                const uint32_t mtype = mat_bits >> 29;
                const float4 mrec = mats16[mat_bits & RT_MAT_INDEX_MASK];
                f3 albedo = mk3(mrec.x, mrec.y, mrec.z);
                const float mparam = mrec.w;

                // EXT: emitted radiance of a diffuse light is added before the scatter decision (accum_radiance of
                // Renderer.cu:157-163); a light never scatters; the last allowed bounce still collects emission.
                bool path_over = false;
                if (EXT) {
                    if (mtype == RT_MAT_DIFFUSE_LIGHT) { accum_rad = accum_rad + atten * albedo; path_over = true; }
                    if (depth + 1u >= p.max_depth) path_over = true;
                }
                RT_PT(9);
                // ---- Scatter (cu_materials.cuh:52-64 / 77-95 / 115-143 / 27-40); per-lane arithmetic is
                // ---- material_scatter()'s, expression by expression.
                const bool is_diel = (mtype == RT_MAT_DIELECTRIC);
                f3 unit_dir = mk3(0.0f);
                float ior_ratio = 0.0f, reflect_prob = 0.0f;
                bool must_reflect = false;
                if (is_diel && !path_over) {
                    bool hit_backface = dot(ray.d, normal) > 0;
                    if (hit_backface) normal = -normal;
                    ior_ratio = hit_backface ? mparam : 1 / mparam;
                    unit_dir = normalize(ray.d);
                    float cos_theta = fminf(dot(-unit_dir, normal), 1.0f);
                    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
                    reflect_prob = reflectance(cos_theta, ior_ratio);
                    must_reflect = ior_ratio * sin_theta > 1.0f;  // short-circuit: no uniform is drawn
                }
                RT_PT(10);
                f3 scatter_dir = mk3(0.0f);
                bool scattered_ok = !path_over;
                if (path_over) {
                    // nothing to sample: the path ends here with what it has collected
                } else if (is_diel) {
                    if (must_reflect || reflect_prob > rng.next()) scatter_dir = reflect(unit_dir, normal);
                    else scatter_dir = refract(unit_dir, normal, ior_ratio);
                } else {
                    RT_PT(11);
                    f3 on_unit = rng_on_unit3(rng);
                    RT_PT(12);
                    if (EXT && mtype == RT_MAT_ISOTROPIC) {
                        scatter_dir = on_unit;   // isotropic phase function: any direction, never absorbed
                    } else if (mtype == RT_MAT_METAL) {
                        scatter_dir = reflect(ray.d, normal) + on_unit * mparam;
                        scattered_ok = !(dot(scatter_dir, normal) < 0 || near_zero(scatter_dir));
                    } else {
                        scatter_dir = normal + on_unit;
                        scattered_ok = !near_zero(scatter_dir);
                        if (mtype == RT_MAT_LAMBERTIAN_CHECKER) {
                            const rt_material& mg = p.scene.mats[mat_bits & RT_MAT_INDEX_MASK];
                            albedo = checker_value(albedo, mk3(mg.albedo2[0], mg.albedo2[1], mg.albedo2[2]), mparam, hit_p);
                        }
                        if (EXT >= 2 && mtype == RT_MAT_LAMBERTIAN_NOISE) albedo = noise_value(p.scene.perlin, albedo, mparam, hit_p);
                        if (EXT >= 2 && mtype == RT_MAT_LAMBERTIAN_IMAGE) {
                            if ((uint32_t)rec_code >= p.scene.sphere_codes) {   // on a quad: (u, v) = the planar coordinates of the hit
                                const float4* qd = quads + ((uint32_t)rec_code - p.scene.sphere_codes) * 4u;
                                const float4 a0 = qd[0], a1 = qd[1], a2 = qd[2], a3 = qd[3];
                                albedo = image_value_quad(p.scene.image, p.scene.image_w, p.scene.image_h, mk3(a0.x, a0.y, a0.z), mk3(a1.x, a1.y, a1.z),
                                                          mk3(a1.w, a2.x, a2.y), mk3(a3.y, a3.z, a3.w), hit_p);
                            } else {
                                albedo = image_value(p.scene.image, p.scene.image_w, p.scene.image_h, normal);
                            }
                        }
                    }
                }
                RT_PT(13);
                if (!scattered_ok) {
                    RT_EMIT_DARK();
                } else {
                    atten = atten * albedo;
                    ray.o = hit_p;
                    ray.d = scatter_dir;  // time is inherited
                    ray.o = ray.o + ray.d * 0.001f;  // Renderer.cu:175
                    depth++;
                    start_trace = true;
                }
            }
        }

        RT_PT(3);
        // ---- hand new samples to the lanes that need one (wave-uniform loop).  The primary ray of sample index n
        // ---- (pixel jitter, camera sample, RNG state after those draws: Renderer.cu:199-201) was computed by
        // ---- primary_rays_kernel with every lane busy; here a lane only loads its 48-byte record.
        for (;;) {
            uint64_t m_need = __ballot(cur == K_NEED);
            if (m_need == 0ull) break;
            if (pool_next == pool_end) {
                if (pool_dry) break;
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(p.work_counter, p.chunk);
                base = __builtin_amdgcn_readfirstlane(base);
                if (base >= p.total) { pool_dry = true; break; }
                pool_next = base;
                pool_end = min(base + p.chunk, p.total);
            }
            uint32_t take = min(pool_end - pool_next, (uint32_t)__popcll(m_need));
            uint32_t rank = lane_rank(m_need);
            if (cur == K_NEED && rank < take) {
                const uint32_t n = pool_next + rank;
                // all three 16-byte parts of the record go out together: ONE global round trip (a padding pixel's two extra loads are wasted, rarely)
                const uint4 rs = RT_LOAD_ONCE(p.prim_rng + n);
                const float4 po = RT_LOAD_ONCE(p.prim_o + n), pd = RT_LOAD_ONCE(p.prim_d + n);
                asm volatile("" : : "v"(po.x), "v"(pd.x));   // keeps the compiler from sinking the two loads below the test (a second, dependent round trip)
                if ((rs.x | rs.y | rs.z | rs.w) != 0u) {
                    out_idx = n;
                    ray.o = mk3(po.x, po.y, po.z); ray.d = mk3(pd.x, pd.y, pd.z); ray.time = po.w;
                    rng.s0 = rs.x; rng.s1 = rs.y; rng.s2 = rs.z; rng.s3 = rs.w;
                    atten = mk3(1.0f);
                    if (EXT) accum_rad = mk3(0.0f);
                    depth = 0;
                    if (p.max_depth == 0u) {
                        RT_EMIT(0.0f, 0.0f, 0.0f);
                    } else {
                        start_trace = true;
                        cur = K_START;  // no longer K_NEED, so the loop does not hand it another sample
                    }
                }
                // a padding pixel (outside the image / past the last tile) consumes the index and the lane asks again
            }
            pool_next += take;
        }
        RT_PT(4);
        if (start_trace) RT_BEGIN_TRACE();
        if (FAST_BVH && __ballot(start_trace && !regular) != 0ull) irr_pending = true;
        if (pool_dry && cur == K_NEED) cur = K_OFF;
        RT_PT(5);
        if (__ballot(cur != K_OFF) == 0ull) break;
    }
#ifdef RT_PHASE_TIMERS
    if (lane == 0)
        for (int i = 0; i < 16; i++) { atomicAdd(p.phase_acc + i, pt_[i]); atomicAdd(p.phase_acc + 16 + i, (unsigned long long)pc_[i]); }
#endif
#undef RT_PT
#undef RT_BEGIN_TRACE
#undef RT_POP
#undef RT_EMIT
#undef RT_EMIT_DARK
}

// Primary rays of one pass: one thread per sample index n = (block * pass_spp + s) * 64 + pixel-in-block (the index space
// the streaming kernel consumes; blockIdx.y = block, so no integer division).  Renderer.cu:188-201: pixel centre in NDC,
// jitter in a disc of half a pixel (cuRandomInUnit<2>), camera sample_ray — with the per-sample counter-based RNG stream
// (Philox seed, then the sequential draws).  Done here, with every lane busy, instead of inside the persistent kernel where
// only the ~21 lanes of a wave whose path just ended would run it (it was 12 % of that kernel's time at one-third lane
// utilisation).  Writes 48 B per sample: (origin, time), (direction, -), RNG state after the camera's draws.
// The two rejection loops of a defocus camera (pixel jitter, lens point: glm_utils.h:84-90 twice) run as ONE loop whose
// iterations serve whichever of the two a lane is at — the same draws in the same order per lane, but the wave iterates
// max(tries_jitter + tries_lens) times instead of max(tries_jitter) + max(tries_lens).
__global__ __launch_bounds__(256) void primary_rays_kernel(StreamParams p, uint32_t first_blk) {
    const uint32_t blk = first_blk + blockIdx.y;
    const uint32_t rem = blockIdx.x * 256u + threadIdx.x;   // index inside the block's 64 * pass_spp samples
    if (rem >= 64u * p.pass_spp) return;
    const uint32_t n = blk * (64u * p.pass_spp) + rem;
    const uint32_t s_local = rem >> 6, pix = rem & 63u;
    uint32_t x0, y0;
    const bool ok = block_origin(p.tm, blk, x0, y0);
    const uint32_t x = x0 + (pix & 7u), y = y0 + (pix >> 3);
    if (!(ok && x < p.width && y < p.height)) {   // padding pixel: outside the image / past the last tile
        p.prim_rng[n] = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    Rng rng;
    rng.init(p.seed, y * p.width + x, p.pass_first_s + s_local, RT_STREAM_RENDER);
    float psx, psy, ndcx, ndcy;
    pixel_ndc(x, y, p.width, p.height, psx, psy, ndcx, ndcy);
    float jx = 0.0f, jy = 0.0f, a = 0.0f, b = 0.0f;
    if (p.cam.type == RT_CAM_DEFOCUS) {
        uint32_t stage = 0;   // 0: pixel jitter (Renderer.cu:199), 1: lens point (cu_Cameras.cuh:55), 2: done
        do {
            const float u = rng.next_signed();
            const float v = rng.next_signed();
            if (length2(u, v) < 1.0f) {
                if (stage == 0u) { jx = u; jy = v; } else { a = u; b = v; }
                stage++;
            }
        } while (stage < 2u);
    } else {
        rng_in_unit2(rng, jx, jy);
        camera_draw(p.cam, rng, a, b);
    }
    const Ray ray = camera_ray(p.cam, ndcx + jx * psx, ndcy + jy * psy, a, b);
    rt_store_stream(p.prim_o + n, make_float4(ray.o.x, ray.o.y, ray.o.z, ray.time));
    rt_store_stream(p.prim_d + n, make_float4(ray.d.x, ray.d.y, ray.d.z, 0.0f));
    rt_store_stream(p.prim_rng + n, make_uint4(rng.s0, rng.s1, rng.s2, rng.s3));
}

// Adds the samples of one pass to each pixel IN SAMPLE ORDER (Renderer.cu:198-204: `radiance += ...`),
// and on the last pass applies mean / clamp / sqrt-gamma / alpha (Renderer.cu:206-216).
__global__ __launch_bounds__(256) void resolve_kernel(StreamParams p, float4* __restrict__ running, float* __restrict__ out, uint32_t last_pass) {
    uint32_t L = blockIdx.x * blockDim.x + threadIdx.x;
    if (L >= p.tm.n_local_tiles * RT_TILE * RT_TILE) return;
    uint32_t gid;
    if (!local_pixel_to_gid(p.tm, L, gid)) return;
    f3 radiance = mk3(0.0f);
    if (p.pass_first_s != 0u) { const float4 r = running[L]; radiance = mk3(r.x, r.y, r.z); }
    size_t src = (size_t)(L >> 6) * p.pass_spp * 64u + (L & 63u);
    for (uint32_t s = 0; s < p.pass_spp; s++) {
        radiance = radiance + load_sample(p.samples, src);
        src += 64u;
    }
    if (last_pass) {
        radiance = radiance * (1.0f / (float)p.spp);
        f3 col = clamp01_sqrt(radiance);
        reinterpret_cast<float4*>(out)[p.tm.direct ? gid : L] = make_float4(col.x, col.y, col.z, 1.0f);
    } else {
        running[L] = make_float4(radiance.x, radiance.y, radiance.z, 0.0f);
    }
}

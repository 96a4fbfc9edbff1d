// rt_xchg_kernel.hpp — render_kernel_xchg: the persistent streaming kernel with rays EXCHANGED between the waves of a workgroup.
//
// render_kernel_stream (rt_stream_kernel.hpp) lets one wave run every phase of its 64 paths — inner-node steps, leaf tests,
// shading, regeneration — one phase at a time, so a lane whose trace has ended waits for the wave's next shade round and a
// lane that is being shaded waits for the wave's next trace: 42 % of the lanes of an issued vector instruction are masked
// off (round 2: SQ_THREAD_CYCLES_VALU / 64 SQ_INSTS_VALU = 0.58).  Rays can only be re-packed into full waves if they can
// move between waves, and the only place a ray is small is the boundary of a trace: there its per-lane traversal stack is
// empty and its state is 18-24 dwords.  So here the waves of a workgroup have two ROLES and hand rays to each other through
// two rings in the LDS:
//
//   tracer waves (0 .. n_tracers-1)  BVH::ClosestIntersection only (BVH.cu:54-106): the hot inner-node loop and the leaf phase of
//       render_kernel_stream, unchanged.  A lane whose trace has ended pushes (ray, closest hit, path state) into the SHADE ring and
//       takes a ready-to-trace ray out of the TRACE ring — a few LDS instructions instead of a shade round of ~400, so the exchange
//       can run as soon as ~16 lanes have finished and the hot loop stays full.
//   shader waves (the rest)          pop 64 finished traces at a time and run sample_world's loop body after the trace
//       (Renderer.cu:149-176: sky / Scatter / depth limit) with EVERY lane busy, hand new samples to the lanes whose path ended,
//       prepare the next trace of every lane (reciprocals, direction signs, root box: BVH.cu:59-60) and push the rays into the TRACE
//       ring.  They also emit the finished samples.
//
// The lane program of a path is still the reference's, expression by expression; only WHICH lane of WHICH wave executes a step has
// changed, and the per-sample result does not depend on that (one RNG stream per sample, per-pixel sums in sample order by
// resolve_kernel): the framebuffer is bit-identical to render_kernel_stream's and the oracle's.
//
// Rings (multi-producer, multi-consumer, all inside one workgroup's LDS; no global memory, no workgroup barrier after start-up):
//   credits   `avail` (entries a consumer may take) and `space` (entries a producer may fill): a wave takes up to n credits with
//             one returning subtract (and gives back what it over-drew), so nobody ever blocks on an empty or full ring;
//   tickets   `head` / `tail`: n consecutive positions per successful reservation;
//   per-slot  sequence words (Vyukov's bounded queue): slot s is free for position p when seq[s] == p and holds position p's entry when
//             seq[s] == p + 1; the reader stores p + capacity.  They order out-of-order completion among concurrent producers /
//             consumers; a wait on them is bounded by another wave's copy of one entry.
//   The LDS executes one wave's instructions in order, so "data, then sequence word" needs no more than a compiler fence restricted
//   to the LDS (s_waitcnt lgkmcnt only: a wave never waits for its global stores here).
// Every spin is bounded: a protocol bug sets an error flag (reported by the host as RT_ERR_HIP), it cannot hang the GPU.
//
// Population: a workgroup keeps `pop_target` rays alive (tracer lanes + what is in flight between the roles).  Only shader waves
// create rays (from the global sample counter) and only they retire them, so a full TRACE ring can never deadlock against a full
// SHADE ring: pop_target < tracer lanes + both capacities + shader lanes.
#pragma once
#include "rt_stream_kernel.hpp"

#define RT_XCHG_BLOCK 768
#define RT_XCHG_SPIN_LIMIT (1u << 22)   // s_sleep(1) rounds (~64 cycles each): ~0.1 s, then the error flag

struct XchgParams {
    StreamParams s;           // scene image, pass, primary rays, sample buffer, work counter: as for render_kernel_stream
    uint32_t n_tracers;       // waves 0 .. n_tracers-1 trace, the others shade
    uint32_t tq_cap, sq_cap;  // ring capacities in entries (powers of two)
    uint32_t pop_target;      // rays kept alive per workgroup
    uint32_t swap_min;        // tracer: exchange once this many lanes are finished or empty
    uint32_t shade_min;       // shader: wait (bounded) until this many finished traces are available
    uint32_t shade_patience;  // ... for at most this many polls
    uint32_t scene_vec4;      // 16-B units of the scene image staged in the LDS: nodes | spheres | extra (when a sphere moves)
    uint32_t extra_in_lds;    // 1: `extra` (second centre, material bits) is part of the LDS image
    uint32_t shader_prio;     // s_setprio of the shader waves
    uint32_t* error_flag;
#ifdef RT_PHASE_TIMERS
    unsigned long long* xphase_acc;   // [0..15] cycles, [16..31] visits
#endif
};

enum : uint32_t { XC_TQ_AVAIL = 0, XC_TQ_SPACE, XC_TQ_HEAD, XC_TQ_TAIL, XC_SQ_AVAIL, XC_SQ_SPACE, XC_SQ_HEAD, XC_SQ_TAIL,
                  XC_POP, XC_DRY, XC_DONE, XC_ERR, XC_WORDS = 16 };

// ring entries, in 16-byte chunks (chunk c of slot s lives at chunk_base[c * capacity + s]: consecutive slots are consecutive
// 16-byte words, so a wave's ds_read/write_b128 of consecutive positions is conflict-free)
//   TRACE ring, 6 chunks: (o, time) (d, dot(d,d)) (RN(1/d), start reference | direction signs << 16 | regular << 19)
//                         (low reciprocal words, depth) (attenuation, sample index) (RNG state)
//   SHADE ring, 4 chunks + 8 bytes: (o, time) (d, closest distance) (attenuation, sample index) (RNG state) | (leaf code, depth)
#define XC_TQ_CHUNKS 6u
#define XC_SQ_CHUNKS 4u
#define XC_TQ_ENTRY_BYTES (XC_TQ_CHUNKS * 16u)
#define XC_SQ_ENTRY_BYTES (XC_SQ_CHUNKS * 16u + 8u)

__device__ __forceinline__ uint32_t xc_ld(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void xc_st(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint32_t xc_add(uint32_t* p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// LDS-only fences: the LDS runs a wave's instructions in order; these keep the COMPILER from moving LDS accesses across and wait
// for the wave's outstanding LDS operations, never for its global-memory traffic
#define XC_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local")
#define XC_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local")

// Take up to `want` credits (never blocks: what is over-drawn is given back) and as many consecutive ring positions.
// Executed by lane 0; the result is wave-uniform.  `credit` is read as a signed number: it is transiently negative while
// several waves over-draw at once, which only makes the others see less than there is.
__device__ __forceinline__ uint32_t xc_reserve(uint32_t* credit, uint32_t* ticket, uint32_t want, uint32_t lane, uint32_t& base) {
    uint32_t got = 0, b = 0;
    if (lane == 0u) {
        if ((int32_t)xc_ld(credit) > 0) {
            const int32_t old = (int32_t)__hip_atomic_fetch_sub(credit, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const int32_t g = old < 0 ? 0 : (old > (int32_t)want ? (int32_t)want : old);
            if (g < (int32_t)want) xc_add(credit, want - (uint32_t)g);
            if (g > 0) b = xc_add(ticket, (uint32_t)g);
            got = (uint32_t)g;
        }
    }
    base = __builtin_amdgcn_readfirstlane(b);
    return __builtin_amdgcn_readfirstlane(got);
}

// wait until the slot's sequence word says `expect`; bounded
__device__ __forceinline__ void xc_wait_seq(const uint32_t* word, uint32_t expect, uint32_t* ctrl, uint32_t* error_flag) {
    uint32_t spins = 0;
    while (xc_ld(word) != expect) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > RT_XCHG_SPIN_LIMIT) {
            xc_st(ctrl + XC_ERR, 1u);
            *error_flag = 1u;
            break;
        }
    }
}

#ifdef RT_PHASE_TIMERS
#define XC_PT(i) do { unsigned long long n_ = __builtin_readcyclecounter(); pt_[i] += n_ - pt_last_; pt_last_ = n_; pc_[i]++; } while (0)
#else
#define XC_PT(i)
#endif

template <int BLOCK>
__global__ __launch_bounds__(BLOCK, BLOCK / 128) void render_kernel_xchg(XchgParams xp) {
    extern __shared__ uint4 lds[];
    const StreamParams& p = xp.s;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;
    constexpr uint32_t K_LEAF = RT_REF_LEAF, K_IRR = RT_REF_IRR, K_SHADE = 0xffffu, K_EMPTY = 0x20000u;
    using ref_t = uint16_t;

    // ---- LDS layout: scene image | per-lane stacks of the tracer waves | control words | sequence words | ring data -------------
    const uint32_t stacks_vec4 = (xp.n_tracers * 64u * p.scene.stack_cap * 2u + 15u) / 16u;
    uint32_t* const ctrl = reinterpret_cast<uint32_t*>(lds + xp.scene_vec4 + stacks_vec4);
    uint32_t* const tq_seq = ctrl + XC_WORDS;
    uint32_t* const sq_seq = tq_seq + xp.tq_cap;
    uint4* const tq_data = reinterpret_cast<uint4*>(sq_seq + xp.sq_cap);          // capacities are multiples of 4: 16-byte aligned
    uint4* const sq_data = tq_data + XC_TQ_CHUNKS * xp.tq_cap;
    uint2* const sq_tail8 = reinterpret_cast<uint2*>(sq_data + XC_SQ_CHUNKS * xp.sq_cap);
    const uint32_t tq_mask = xp.tq_cap - 1u, sq_mask = xp.sq_cap - 1u;

    for (uint32_t i = tid; i < xp.scene_vec4; i += BLOCK) lds[i] = p.scene.blob[i];
    if (tid < XC_WORDS) ctrl[tid] = tid == XC_TQ_SPACE ? xp.tq_cap : (tid == XC_SQ_SPACE ? xp.sq_cap : 0u);
    for (uint32_t i = tid; i < xp.tq_cap; i += BLOCK) tq_seq[i] = i;
    for (uint32_t i = tid; i < xp.sq_cap; i += BLOCK) sq_seq[i] = i;
    __syncthreads();

    const char* nodes = reinterpret_cast<const char*>(lds);
    const float4* spheres = reinterpret_cast<const float4*>(lds + p.scene.off_spheres);
    // second centres + material bits: in the LDS image when a sphere moves (the leaf phase reads them), else from global memory / L1
    const float4* extra = xp.extra_in_lds ? reinterpret_cast<const float4*>(lds + p.scene.off_extra)
                                          : reinterpret_cast<const float4*>(p.scene.blob + p.scene.off_extra);
    const float4* mats16 = reinterpret_cast<const float4*>(p.scene.blob + p.scene.off_mats);   // shader waves only: global memory / L1

#ifdef RT_PHASE_TIMERS
    unsigned long long pt_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pt_last_ = __builtin_readcyclecounter();
    uint32_t pc_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif

    if (wave < xp.n_tracers) {
        // =====================================================================================================================
        // tracer wave
        // =====================================================================================================================
        ref_t* const stack = reinterpret_cast<ref_t*>(lds + xp.scene_vec4) + wave * 64u * p.scene.stack_cap + lane;
        *stack = (ref_t)K_SHADE;   // entry 0: popping an empty stack IS "trace finished"
        Ray ray;
        ray.o = mk3(0.0f); ray.d = mk3(0.0f); ray.time = 0.0f;
        f3 inv_d = mk3(0.0f), inv_lo = mk3(0.0f);
        float ray_a = 0.0f, rec_t = RT_MISS_DIST;
        int32_t rec_code = -1;
        uint32_t cur = K_EMPTY;
        ref_t* sp = stack + 64;
        uint32_t kx = 0, ky = 0, kz = 0;
        bool regular = true, irr_pending = false;
        // path state that only travels with the ray
        float4 c_att = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // attenuation, sample index
        uint4 c_rng = make_uint4(0u, 0u, 0u, 0u);
        uint32_t depth = 0;

        for (;;) {
            // ---------------- phase 1: inner-node steps (BVH.cu:76-97), as render_kernel_stream's FAST_BVH hot loop ----------------
            bool at_inner = cur < K_IRR;
            if (__ballot(at_inner) != 0ull) {
                uint32_t n_inner_lanes;
                do {
                    if (at_inner) {
                        const WideNodeData nd = fetch_wide_node<false>(nodes, lds, 0u, cur, kx, ky, kz);
                        const uint32_t left_idx = nd.left, right_idx = nd.right;
                        float tl, tr;
                        const bool hl = slab_near_far_regular(nd.lnx, nd.lny, nd.lnz, nd.lfx, nd.lfy, nd.lfz, ray, inv_d, inv_lo, rec_t, tl);
                        const bool hr = slab_near_far_regular(nd.rnx, nd.rny, nd.rnz, nd.rfx, nd.rfy, nd.rfz, ray, inv_d, inv_lo, rec_t, tr);
                        const bool go_right = hr && (!hl || tl > tr);
                        if (hl && hr) {
                            *sp = (ref_t)(go_right ? left_idx : right_idx);
                            sp += 64;
                        }
                        cur = go_right ? right_idx : left_idx;
                        if (!(hl || hr)) { sp -= 64; cur = *sp; }
                    }
                    at_inner = cur < K_IRR;
                    n_inner_lanes = (uint32_t)__popcll(__ballot(at_inner));
                } while (n_inner_lanes >= p.inner_keep);
            }
            XC_PT(0);
            if (irr_pending) {   // wave-uniform, rare: rays outside the fast-division class (marked references), verbatim box tests
                for (;;) {
                    const bool at_irr = (cur & (K_LEAF | K_IRR)) == K_IRR;
                    if (__ballot(at_irr) == 0ull) break;
                    if (at_irr) {
                        const WideNodeData nd = fetch_wide_node<false>(nodes, lds, 0u, cur & (K_IRR - 1u), 0u, 0u, 0u);
                        uint32_t left_idx = nd.left, right_idx = nd.right;
                        if (left_idx < K_LEAF) left_idx |= K_IRR;
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
                        if (!(hl || hr)) { sp -= 64; cur = *sp; }
                    }
                }
                irr_pending = __ballot(!regular && (cur < K_SHADE)) != 0ull;
            }
            XC_PT(1);
            // ---------------- phase 2: leaves (BVH.cu:69-73 -> SphereHittable.cu:56-66 / :91-102) ----------------------------------
            {
                const bool at_leaf = (cur - K_LEAF) < (K_SHADE - K_LEAF);
                const uint64_t m_leaf = __ballot(at_leaf);
                if (m_leaf != 0ull && ((uint32_t)__popcll(m_leaf) >= p.leaf_min || __ballot(cur < K_LEAF) == 0ull)) {
                    if (at_leaf) {
                        const uint32_t code = cur & (K_LEAF - 1u);   // prim * 2 + is_moving
                        const uint32_t prim = code >> 1;
                        const float4 sph = spheres[prim];
                        f3 center = mk3(sph.x, sph.y, sph.z);
                        if (code & 1u) {
                            const float4 ex = extra[prim];
                            center = mix(center, mk3(ex.x, ex.y, ex.z), ray.time);
                        }
                        const float t = sphere_closest_intersection_a(ray, ray_a, center, sph.w);
                        if (!(t >= rec_t)) {  // `if (t >= rec.distance) return false;`
                            rec_t = t;
                            rec_code = (int32_t)code;
                        }
                        sp -= 64;
                        cur = *sp;
                    }
                }
            }
            XC_PT(2);
            // ---------------- phase 3: exchange finished traces for fresh rays -------------------------------------------------------
            const uint64_t m_trav = __ballot(cur < K_SHADE);
            const uint64_t m_fin = __ballot(cur == K_SHADE);
            const uint32_t n_fin = (uint32_t)__popcll(m_fin);
            const uint32_t n_idle = 64u - (uint32_t)__popcll(m_trav);
            if (n_idle < xp.swap_min && m_trav != 0ull) continue;

            if (n_fin != 0u) {   // finished traces -> SHADE ring
                uint32_t base;
                const uint32_t got = xc_reserve(ctrl + XC_SQ_SPACE, ctrl + XC_SQ_TAIL, n_fin, lane, base);
                const uint32_t rank = lane_rank(m_fin);
                if (cur == K_SHADE && rank < got) {
                    const uint32_t pos = base + rank, slot = pos & sq_mask;
                    xc_wait_seq(sq_seq + slot, pos, ctrl, xp.error_flag);
                    XC_ACQUIRE();
                    sq_data[slot] = make_uint4(__float_as_uint(ray.o.x), __float_as_uint(ray.o.y), __float_as_uint(ray.o.z), __float_as_uint(ray.time));
                    sq_data[xp.sq_cap + slot] = make_uint4(__float_as_uint(ray.d.x), __float_as_uint(ray.d.y), __float_as_uint(ray.d.z), __float_as_uint(rec_t));
                    sq_data[2u * xp.sq_cap + slot] = make_uint4(__float_as_uint(c_att.x), __float_as_uint(c_att.y), __float_as_uint(c_att.z), __float_as_uint(c_att.w));
                    sq_data[3u * xp.sq_cap + slot] = c_rng;
                    sq_tail8[slot] = make_uint2((uint32_t)rec_code, depth);
                    XC_RELEASE();
                    xc_st(sq_seq + slot, pos + 1u);
                    cur = K_EMPTY;
                }
                if (got != 0u && lane == 0u) { XC_RELEASE(); xc_add(ctrl + XC_SQ_AVAIL, got); }
            }
            const uint64_t m_empty = __ballot(cur == K_EMPTY);
            uint32_t n_new = 0;
            if (m_empty != 0ull) {   // fresh rays <- TRACE ring
                uint32_t base;
                n_new = xc_reserve(ctrl + XC_TQ_AVAIL, ctrl + XC_TQ_HEAD, (uint32_t)__popcll(m_empty), lane, base);
                const uint32_t rank = lane_rank(m_empty);
                if (cur == K_EMPTY && rank < n_new) {
                    const uint32_t pos = base + rank, slot = pos & tq_mask;
                    xc_wait_seq(tq_seq + slot, pos + 1u, ctrl, xp.error_flag);
                    XC_ACQUIRE();
                    const uint4 a0 = tq_data[slot], a1 = tq_data[xp.tq_cap + slot], a2 = tq_data[2u * xp.tq_cap + slot];
                    const uint4 a3 = tq_data[3u * xp.tq_cap + slot], a4 = tq_data[4u * xp.tq_cap + slot], a5 = tq_data[5u * xp.tq_cap + slot];
                    XC_RELEASE();   // the reads have returned before the slot is handed back
                    xc_st(tq_seq + slot, pos + xp.tq_cap);
                    ray.o = mk3(__uint_as_float(a0.x), __uint_as_float(a0.y), __uint_as_float(a0.z)); ray.time = __uint_as_float(a0.w);
                    ray.d = mk3(__uint_as_float(a1.x), __uint_as_float(a1.y), __uint_as_float(a1.z)); ray_a = __uint_as_float(a1.w);
                    inv_d = mk3(__uint_as_float(a2.x), __uint_as_float(a2.y), __uint_as_float(a2.z));
                    inv_lo = mk3(__uint_as_float(a3.x), __uint_as_float(a3.y), __uint_as_float(a3.z));
                    depth = a3.w;
                    c_att = make_float4(__uint_as_float(a4.x), __uint_as_float(a4.y), __uint_as_float(a4.z), __uint_as_float(a4.w));
                    c_rng = a5;
                    cur = a2.w & 0xffffu;
                    kx = (a2.w >> 14) & 4u; ky = (a2.w >> 15) & 4u; kz = (a2.w >> 16) & 4u;
                    regular = ((a2.w >> 19) & 1u) != 0u;
                    rec_t = RT_MISS_DIST;
                    rec_code = -1;
                    sp = stack + 64;
                }
                if (n_new != 0u && lane == 0u) { XC_RELEASE(); xc_add(ctrl + XC_TQ_SPACE, n_new); }
                if (__ballot(!regular && cur < K_SHADE) != 0ull) irr_pending = true;
            }
            XC_PT(3);
            if (__ballot(cur < K_SHADE) == 0ull && n_new == 0u) {   // nothing to trace and nothing came: the end, or the shaders are behind
                if (xc_ld(ctrl + XC_DONE) != 0u || xc_ld(ctrl + XC_ERR) != 0u) break;
                __builtin_amdgcn_s_sleep(8);
                XC_PT(4);
            }
        }
    } else {
        // =====================================================================================================================
        // shader wave
        // =====================================================================================================================
        if (xp.shader_prio) __builtin_amdgcn_s_setprio(1);
        const uint32_t n_shaders = BLOCK / 64u - xp.n_tracers;
        uint32_t pool_next = 0, pool_end = 0;
        bool pool_dry = false;   // this wave's pool is empty and the global sample counter is exhausted
        const f3 root_min = mk3(p.scene.root_min[0], p.scene.root_min[1], p.scene.root_min[2]);
        const f3 root_max = mk3(p.scene.root_max[0], p.scene.root_max[1], p.scene.root_max[2]);
        enum : uint32_t { S_NONE = 0, S_RAY = 1, S_NEED = 2 };

        for (;;) {
            // ---------------- (1) wait for work: finished traces to shade, or room in the population for new paths ---------------------
            uint32_t tries = 0;
            bool quit = false;
            for (;;) {
                const int32_t avail = (int32_t)xc_ld(ctrl + XC_SQ_AVAIL);
                if (avail >= (int32_t)xp.shade_min || (avail > 0 && tries >= xp.shade_patience)) break;
                const uint32_t pop = xc_ld(ctrl + XC_POP);
                if (avail <= 0 && !pool_dry && pop < xp.pop_target) break;          // start new paths
                if (xc_ld(ctrl + XC_DONE) != 0u || xc_ld(ctrl + XC_ERR) != 0u) { quit = true; break; }
                if (avail <= 0 && pop == 0u && xc_ld(ctrl + XC_DRY) == n_shaders) {  // no ray alive anywhere and nobody can create one
                    xc_st(ctrl + XC_DONE, 1u);
                    quit = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
                tries++;
            }
            if (quit) break;
            XC_PT(8);

            uint32_t base = 0;
            const uint32_t got = xc_reserve(ctrl + XC_SQ_AVAIL, ctrl + XC_SQ_HEAD, 64u, lane, base);
            Ray ray;
            ray.o = mk3(0.0f); ray.d = mk3(0.0f); ray.time = 0.0f;
            f3 atten = mk3(0.0f);
            Rng rng;
            rng.s0 = 1u; rng.s1 = 0u; rng.s2 = 0u; rng.s3 = 0u; rng.draws = 0u;
            float rec_t = RT_MISS_DIST;
            int32_t rec_code = -1;
            uint32_t depth = 0, out_idx = 0;
            uint32_t state = S_NONE;
            if (lane < got) {
                const uint32_t pos = base + lane, slot = pos & sq_mask;
                xc_wait_seq(sq_seq + slot, pos + 1u, ctrl, xp.error_flag);
                XC_ACQUIRE();
                const uint4 a0 = sq_data[slot], a1 = sq_data[xp.sq_cap + slot], a2 = sq_data[2u * xp.sq_cap + slot], a3 = sq_data[3u * xp.sq_cap + slot];
                const uint2 a4 = sq_tail8[slot];
                XC_RELEASE();
                xc_st(sq_seq + slot, pos + xp.sq_cap);
                ray.o = mk3(__uint_as_float(a0.x), __uint_as_float(a0.y), __uint_as_float(a0.z)); ray.time = __uint_as_float(a0.w);
                ray.d = mk3(__uint_as_float(a1.x), __uint_as_float(a1.y), __uint_as_float(a1.z)); rec_t = __uint_as_float(a1.w);
                atten = mk3(__uint_as_float(a2.x), __uint_as_float(a2.y), __uint_as_float(a2.z)); out_idx = a2.w;
                rng.s0 = a3.x; rng.s1 = a3.y; rng.s2 = a3.z; rng.s3 = a3.w;
                rec_code = (int32_t)a4.x; depth = a4.y;
            }
            if (got != 0u && lane == 0u) { XC_RELEASE(); xc_add(ctrl + XC_SQ_SPACE, got); }
            XC_PT(9);

            // ---------------- (2) sample_world's loop body after the trace (Renderer.cu:149-176) ------------------------------------
            if (lane < got) {
                if (rec_code < 0) {
                    const float t = normalize(ray.d).y * 0.5f + 0.5f;
                    const f3 sky = linear_interpolate(mk3(0.1f, 0.2f, 0.4f), mk3(0.9f, 0.9f, 0.99f), t);
                    const f3 rad = atten * sky;
                    p.samples[out_idx] = make_float4(rad.x, rad.y, rad.z, 0.0f);
                    state = S_NEED;
                } else if (depth + 1u >= p.max_depth) {
                    p.samples[out_idx] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // the scatter of the last allowed bounce cannot reach the sky
                    state = S_NEED;
                } else {
                    const f3 hit_p = ray_at(ray, rec_t);
                    const uint32_t prim = (uint32_t)rec_code >> 1;
                    const float4 sph = spheres[prim];
                    const float4 ex = extra[prim];
                    f3 center = mk3(sph.x, sph.y, sph.z);
                    if ((uint32_t)rec_code & 1u) center = mix(center, mk3(ex.x, ex.y, ex.z), ray.time);
                    f3 normal = (hit_p - center) / sph.w;  // SphereHittable.cu:64 / :100
                    const uint32_t mat_bits = __float_as_uint(ex.w);
                    const uint32_t mtype = mat_bits >> 29;
                    const float4 mrec = mats16[mat_bits & RT_MAT_INDEX_MASK];
                    f3 albedo = mk3(mrec.x, mrec.y, mrec.z);
                    const float mparam = mrec.w;
                    // Scatter (cu_materials.cuh:52-64 / 77-95 / 115-143 / 27-40), material_scatter()'s arithmetic expression by expression
                    const bool is_diel = (mtype == RT_MAT_DIELECTRIC);
                    f3 unit_dir = mk3(0.0f);
                    float ior_ratio = 0.0f, reflect_prob = 0.0f;
                    bool must_reflect = false;
                    if (is_diel) {
                        const bool hit_backface = dot(ray.d, normal) > 0;
                        if (hit_backface) normal = -normal;
                        ior_ratio = hit_backface ? mparam : 1 / mparam;
                        unit_dir = normalize(ray.d);
                        const float cos_theta = fminf(dot(-unit_dir, normal), 1.0f);
                        const float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
                        reflect_prob = reflectance(cos_theta, ior_ratio);
                        must_reflect = ior_ratio * sin_theta > 1.0f;  // short-circuit: no uniform is drawn
                    }
                    f3 scatter_dir = mk3(0.0f);
                    bool scattered_ok = true;
                    if (is_diel) {
                        if (must_reflect || reflect_prob > rng.next()) scatter_dir = reflect(unit_dir, normal);
                        else scatter_dir = refract(unit_dir, normal, ior_ratio);
                    } else {
                        const f3 on_unit = rng_on_unit3(rng);
                        if (mtype == RT_MAT_METAL) {
                            scatter_dir = reflect(ray.d, normal) + on_unit * mparam;
                            scattered_ok = !(dot(scatter_dir, normal) < 0 || near_zero(scatter_dir));
                        } else {
                            scatter_dir = normal + on_unit;
                            scattered_ok = !near_zero(scatter_dir);
                            if (mtype == RT_MAT_LAMBERTIAN_CHECKER) {
                                const rt_material& mg = p.scene.mats[mat_bits & RT_MAT_INDEX_MASK];
                                albedo = checker_value(albedo, mk3(mg.albedo2[0], mg.albedo2[1], mg.albedo2[2]), mparam, hit_p);
                            }
                        }
                    }
                    if (!scattered_ok) {
                        p.samples[out_idx] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        state = S_NEED;
                    } else {
                        atten = atten * albedo;
                        ray.o = hit_p;
                        ray.d = scatter_dir;  // time is inherited
                        ray.o = ray.o + ray.d * 0.001f;  // Renderer.cu:175
                        depth++;
                        state = S_RAY;
                    }
                }
            }
            XC_PT(10);

            // ---------------- (3) new paths: for the lanes whose path ended, and — while the population is below target — for idle lanes ----
            if (!pool_dry && got < 64u) {
                const uint32_t n_free = 64u - got;
                uint32_t allowed = 0;
                if (lane == 0u) {
                    if (xc_ld(ctrl + XC_POP) < xp.pop_target) {
                        const uint32_t old = xc_add(ctrl + XC_POP, n_free);
                        allowed = old >= xp.pop_target ? 0u : min(n_free, xp.pop_target - old);
                        if (allowed < n_free) __hip_atomic_fetch_sub(ctrl + XC_POP, n_free - allowed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                allowed = __builtin_amdgcn_readfirstlane(allowed);
                if (lane >= got && lane - got < allowed) state = S_NEED;
            }
            for (;;) {
                const uint64_t m_need = __ballot(state == S_NEED);
                if (m_need == 0ull) break;
                if (pool_next == pool_end) {
                    if (pool_dry) break;
                    uint32_t b = 0;
                    if (lane == 0u) b = atomicAdd(p.work_counter, p.chunk);
                    b = __builtin_amdgcn_readfirstlane(b);
                    if (b >= p.total) {
                        pool_dry = true;
                        if (lane == 0u) xc_add(ctrl + XC_DRY, 1u);
                        break;
                    }
                    pool_next = b;
                    pool_end = min(b + p.chunk, p.total);
                }
                const uint32_t take = min(pool_end - pool_next, (uint32_t)__popcll(m_need));
                const uint32_t rank = lane_rank(m_need);
                if (state == S_NEED && rank < take) {
                    const uint32_t n = pool_next + rank;
                    const uint4 rs = p.prim_rng[n];
                    if ((rs.x | rs.y | rs.z | rs.w) != 0u) {
                        const float4 po = p.prim_o[n], pd = p.prim_d[n];
                        out_idx = n;
                        ray.o = mk3(po.x, po.y, po.z); ray.d = mk3(pd.x, pd.y, pd.z); ray.time = po.w;
                        rng.s0 = rs.x; rng.s1 = rs.y; rng.s2 = rs.z; rng.s3 = rs.w;
                        atten = mk3(1.0f);
                        depth = 0;
                        if (p.max_depth == 0u) p.samples[n] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // and the lane asks again
                        else state = S_RAY;
                    }
                    // a padding pixel (outside the image / past the last tile) consumes the index and the lane asks again
                }
                pool_next += take;
            }
            {   // lanes that wanted a sample and found the queue dry: their paths (ended, or only reserved) leave the population
                const uint32_t unserved = (uint32_t)__popcll(__ballot(state == S_NEED));
                if (unserved != 0u && lane == 0u) __hip_atomic_fetch_sub(ctrl + XC_POP, unserved, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            XC_PT(11);

            // ---------------- (4) prepare the trace (BVH.cu:59-60: root box first, against rec.distance = _MISS_DIST) --------------------
            uint32_t word = 0;
            f3 inv_d = mk3(0.0f), inv_lo = mk3(0.0f);
            float ray_a = 0.0f;
            if (state == S_RAY) {
                ray_a = dot(ray.d, ray.d);
                const bool regular = ray_is_regular(ray);
                inv_d = mk3(rcp_exact_regular(ray.d.x), rcp_exact_regular(ray.d.y), rcp_exact_regular(ray.d.z));   // used by regular rays only
                inv_lo = mk3(rcp_low_word(ray.d.x, inv_d.x), rcp_low_word(ray.d.y, inv_d.y), rcp_low_word(ray.d.z, inv_d.z));
                float d_root;
                const bool hit_root = regular ? aabb_intersects_regular(root_min, root_max, ray, inv_d, RT_MISS_DIST, d_root)
                                              : aabb_intersects(root_min, root_max, ray, RT_MISS_DIST, d_root);
                uint32_t cur0 = K_SHADE;
                if (hit_root) {
                    cur0 = p.scene.root_ref;
                    if (!regular && cur0 < K_LEAF) cur0 |= K_IRR;
                }
                // direction signs select the (near, far) plane pair by address in the tracer (regular rays only)
                const uint32_t sx = regular ? (__float_as_uint(ray.d.x) >> 31) : 0u, sy = regular ? (__float_as_uint(ray.d.y) >> 31) : 0u,
                               sz = regular ? (__float_as_uint(ray.d.z) >> 31) : 0u;
                word = cur0 | (sx << 16) | (sy << 17) | (sz << 18) | ((regular ? 1u : 0u) << 19);
            }
            XC_PT(12);

            // ---------------- (5) rays -> TRACE ring (waits while the ring is full: the tracers always drain it) -------------------------
            {
                const uint64_t m_ray = __ballot(state == S_RAY);
                const uint32_t n_ray = (uint32_t)__popcll(m_ray);
                const uint32_t rank = lane_rank(m_ray);
                uint32_t pushed = 0, spins = 0;
                while (pushed < n_ray) {
                    uint32_t tb;
                    const uint32_t g = xc_reserve(ctrl + XC_TQ_SPACE, ctrl + XC_TQ_TAIL, n_ray - pushed, lane, tb);
                    if (g == 0u) {
                        __builtin_amdgcn_s_sleep(2);
                        if (xc_ld(ctrl + XC_ERR) != 0u) break;
                        if (++spins > RT_XCHG_SPIN_LIMIT) { xc_st(ctrl + XC_ERR, 1u); *xp.error_flag = 1u; break; }
                        continue;
                    }
                    if (state == S_RAY && rank >= pushed && rank < pushed + g) {
                        const uint32_t pos = tb + (rank - pushed), slot = pos & tq_mask;
                        xc_wait_seq(tq_seq + slot, pos, ctrl, xp.error_flag);
                        XC_ACQUIRE();
                        tq_data[slot] = make_uint4(__float_as_uint(ray.o.x), __float_as_uint(ray.o.y), __float_as_uint(ray.o.z), __float_as_uint(ray.time));
                        tq_data[xp.tq_cap + slot] = make_uint4(__float_as_uint(ray.d.x), __float_as_uint(ray.d.y), __float_as_uint(ray.d.z), __float_as_uint(ray_a));
                        tq_data[2u * xp.tq_cap + slot] = make_uint4(__float_as_uint(inv_d.x), __float_as_uint(inv_d.y), __float_as_uint(inv_d.z), word);
                        tq_data[3u * xp.tq_cap + slot] = make_uint4(__float_as_uint(inv_lo.x), __float_as_uint(inv_lo.y), __float_as_uint(inv_lo.z), depth);
                        tq_data[4u * xp.tq_cap + slot] = make_uint4(__float_as_uint(atten.x), __float_as_uint(atten.y), __float_as_uint(atten.z), out_idx);
                        tq_data[5u * xp.tq_cap + slot] = make_uint4(rng.s0, rng.s1, rng.s2, rng.s3);
                        XC_RELEASE();
                        xc_st(tq_seq + slot, pos + 1u);
                    }
                    if (lane == 0u) { XC_RELEASE(); xc_add(ctrl + XC_TQ_AVAIL, g); }
                    pushed += g;
                }
            }
            XC_PT(13);
        }
    }
#ifdef RT_PHASE_TIMERS
    if (lane == 0)
        for (int i = 0; i < 16; i++) { atomicAdd(xp.xphase_acc + i, pt_[i]); atomicAdd(xp.xphase_acc + 16 + i, (unsigned long long)pc_[i]); }
#endif
}
#undef XC_PT

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
// Rings (multi-producer, multi-consumer, all inside one workgroup's LDS; no global memory, no workgroup barrier after start-up).
// Every synchronising step is an LDS round trip of a few hundred cycles under load, so the protocol is built to need few of them:
//   ONE 64-bit control word per ring pair: (shade head, shade tail, trace head, trace tail), 16 bits each, positions modulo 2^16.
//             A tracer reserves its pushes (shade tail += n) AND its pops (trace head += m) with ONE ds_cmpst_rtn_b64 — it takes
//             what is there (never waits for an empty or a full ring); a shader pops (shade head) or pushes (trace tail) with one.
//   per-slot  sequence words (Vyukov's bounded queue): slot s is free for position p when seq[s] == p and holds position p's entry when
//             seq[s] == p + 1; the reader stores p + capacity.  They order out-of-order completion among concurrent producers /
//             consumers; a wait on them is bounded by another wave's copy of one entry.  They are loaded speculatively TOGETHER with
//             the compare-and-swap (for the positions it will grant if it succeeds), not after it.
//   The LDS executes one wave's instructions in order, so "data, then sequence word" (and "data read, then slot release") need only a
//   COMPILER fence (wavefront scope, LDS address space: no s_waitcnt); a wave never waits for its global stores here.
//   An exchange of a tracer is then: load the word, compare-and-swap (+ sequence words), entry reads = 3 round trips.
//   A workgroup may run several independent ring pairs (`shards`: tracers / shaders split evenly), which divides the traffic on a word.
// Every spin is bounded: a protocol bug sets an error flag (reported by the host as RT_ERR_HIP), it cannot hang the GPU.
//
// Population: a ring pair keeps `pop_target` rays alive (its tracers' lanes + what is in flight between the roles).  Only shader waves
// create rays (from the global sample counter) and only they retire them, so a full TRACE ring can never deadlock against a full
// SHADE ring: pop_target < tracer lanes + both capacities + shader lanes.
#pragma once
#include "rt_stream_kernel.hpp"

#define RT_XCHG_SPIN_LIMIT (1u << 16)   // rounds of s_sleep + one LDS read (~400 cycles): ~10 ms, then the error flag
#define RT_XCHG_IDLE_LIMIT (1u << 19)   // idle rounds of a wave that waits for the other role (~600 cycles each): ~0.1 s

struct XchgParams {
    StreamParams s;           // scene image, pass, primary rays, sample buffer, work counter: as for render_kernel_stream
    uint32_t n_tracers;       // waves 0 .. n_tracers-1 trace, the others shade
    uint32_t n_shards;        // independent ring pairs per workgroup (1 or 2)
    uint32_t tq_cap, sq_cap;  // ring capacities in entries PER SHARD (powers of two)
    uint32_t pop_extra;       // rays kept alive per workgroup beyond the tracer lanes (split over the shards)
    uint32_t swap_min;        // tracer: exchange once this many lanes are finished or empty
    uint32_t shade_min;       // shader: wait (bounded) until this many finished traces are available
    uint32_t shade_patience;  // ... for at most this many polls
    uint32_t scene_vec4;      // 16-B units of the scene image staged in the LDS: nodes | spheres | extra (when a sphere moves)
    uint32_t extra_in_lds;    // 1: `extra` (second centre, material bits) is part of the LDS image
    uint32_t shader_prio;     // s_setprio of the shader waves
    uint32_t* error_flag;     // [0] flag; [16 ...] RT_XCHG_DEBUG_WORDS per wave of the grid, written by a wave that gave up
#ifdef RT_PHASE_TIMERS
    unsigned long long* xphase_acc;   // [0..15] cycles, [16..31] visits
#endif
};

// control block of a shard: 16 dwords; [0..1] = the 64-bit ring word
enum : uint32_t { XC_WORD = 0, XC_POP = 2, XC_DRY, XC_DONE, XC_ERR, XC_WORDS = 16 };

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
__device__ __forceinline__ uint64_t xc_uniform64(uint64_t v) {
    // (the builtin returns a signed int: without the casts the low half would be SIGN-extended over the high half)
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);
}
// the ring word as every lane sees it (one broadcast read), made wave-uniform
__device__ __forceinline__ uint64_t xc_word(const uint64_t* w) { return xc_uniform64(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); }
// compare-and-swap by lane 0; returns the value found (== expected on success), wave-uniform
__device__ __forceinline__ uint64_t xc_cas(uint64_t* w, uint64_t expected, uint64_t desired, uint32_t lane) {
    uint64_t found = expected;
    if (lane == 0u) __hip_atomic_compare_exchange_strong(w, &found, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return xc_uniform64(found);
}
#define XC_SQ_H(w) ((uint32_t)(w) & 0xffffu)
#define XC_SQ_T(w) (((uint32_t)(w) >> 16) & 0xffffu)
#define XC_TQ_H(w) ((uint32_t)((w) >> 32) & 0xffffu)
#define XC_TQ_T(w) ((uint32_t)((w) >> 48) & 0xffffu)
__device__ __forceinline__ uint64_t xc_pack(uint32_t sq_h, uint32_t sq_t, uint32_t tq_h, uint32_t tq_t) {
    return (uint64_t)((sq_h & 0xffffu) | ((sq_t & 0xffffu) << 16)) | ((uint64_t)((tq_h & 0xffffu) | ((tq_t & 0xffffu) << 16)) << 32);
}
// LDS-only fences.  RELEASE waits for the wave's outstanding LDS operations (s_waitcnt lgkmcnt(0), never vmcnt) before the sequence
// word is stored: "entry, then sequence word" and "entry read, then slot handed back".  ACQUIRE is compiler-only: the sequence word's
// value has been tested, so its load has returned before the entry is read.
#define XC_ORDER_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local")
#define XC_ORDER_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local")

// wait until the low 16 bits of the slot's sequence word are `expect` (`have` = the value already loaded); bounded
__device__ __forceinline__ void xc_wait_seq(const uint32_t* word, uint32_t have, uint32_t expect, uint32_t* ctrl, uint32_t* error_flag) {
    uint32_t spins = 0;
    while (((have ^ expect) & 0xffffu) != 0u) {
        __builtin_amdgcn_s_sleep(1);
        have = xc_ld(word);
        if (++spins > RT_XCHG_SPIN_LIMIT || ((spins & 255u) == 0u && xc_ld(ctrl + XC_ERR) != 0u)) {
            xc_st(ctrl + XC_ERR, 2u);
            *error_flag = 2u;
            break;
        }
    }
}
// what a wave that gave up saw (diagnostics of a protocol failure: never expected)
__device__ __forceinline__ void xc_dump(uint32_t* error_flag, uint32_t why, uint32_t wave, uint32_t lane, const uint32_t* ctrl, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    if (lane != 0u) return;
    uint32_t* o = error_flag + 16u + (blockIdx.x * (blockDim.x / 64u) + wave) * RT_XCHG_DEBUG_WORDS;
    o[0] = why; o[1] = a; o[2] = b; o[3] = c; o[4] = d;
    o[5] = xc_ld(ctrl + XC_WORD); o[6] = xc_ld(ctrl + XC_WORD + 1); o[7] = xc_ld(ctrl + XC_POP); o[8] = xc_ld(ctrl + XC_DRY);
    o[9] = xc_ld(ctrl + XC_DONE); o[10] = xc_ld(ctrl + XC_ERR);
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

    // ---- LDS layout: scene image | per-lane stacks of the tracer waves | per shard: control words, sequence words | ring data ----
    const uint32_t n_waves = BLOCK / 64u, n_shaders_all = n_waves - xp.n_tracers;
    const bool is_tracer = wave < xp.n_tracers;
    // shard of this wave, and who else is in it
    const uint32_t shard = is_tracer ? wave * xp.n_shards / xp.n_tracers : (wave - xp.n_tracers) * xp.n_shards / n_shaders_all;
    uint32_t shard_tracers = 0, shard_shaders = 0;
    for (uint32_t k = 0; k < xp.n_tracers; k++) shard_tracers += (k * xp.n_shards / xp.n_tracers == shard) ? 1u : 0u;
    for (uint32_t k = 0; k < n_shaders_all; k++) shard_shaders += (k * xp.n_shards / n_shaders_all == shard) ? 1u : 0u;
    const uint32_t pop_target = shard_tracers * 64u + xp.pop_extra / xp.n_shards;
    const uint32_t stacks_vec4 = (xp.n_tracers * 64u * p.scene.stack_cap * 2u + 15u) / 16u;
    uint32_t* const ctrl_all = reinterpret_cast<uint32_t*>(lds + xp.scene_vec4 + stacks_vec4);
    uint32_t* const ctrl = ctrl_all + shard * XC_WORDS;
    uint64_t* const ring_word = reinterpret_cast<uint64_t*>(ctrl + XC_WORD);
    uint32_t* const seq_all = ctrl_all + xp.n_shards * XC_WORDS;
    uint32_t* const tq_seq = seq_all + shard * (xp.tq_cap + xp.sq_cap);
    uint32_t* const sq_seq = tq_seq + xp.tq_cap;
    uint4* const data_all = reinterpret_cast<uint4*>(seq_all + xp.n_shards * (xp.tq_cap + xp.sq_cap));   // capacities are multiples of 4: 16-byte aligned
    uint4* const tq_data = data_all + shard * (XC_TQ_CHUNKS * xp.tq_cap);
    uint4* const sq_data = data_all + xp.n_shards * (XC_TQ_CHUNKS * xp.tq_cap) + shard * (XC_SQ_CHUNKS * xp.sq_cap);
    uint2* const sq_tail8 = reinterpret_cast<uint2*>(data_all + xp.n_shards * (XC_TQ_CHUNKS * xp.tq_cap + XC_SQ_CHUNKS * xp.sq_cap)) + shard * xp.sq_cap;
    const uint32_t tq_mask = xp.tq_cap - 1u, sq_mask = xp.sq_cap - 1u;

    for (uint32_t i = tid; i < xp.scene_vec4; i += BLOCK) lds[i] = p.scene.blob[i];
    if (tid < xp.n_shards * XC_WORDS) ctrl_all[tid] = 0u;
    for (uint32_t i = tid; i < xp.n_shards * (xp.tq_cap + xp.sq_cap); i += BLOCK) {   // seq[s] = s: slot s is free for position s
        const uint32_t k = i % (xp.tq_cap + xp.sq_cap);
        seq_all[i] = k < xp.tq_cap ? k : k - xp.tq_cap;
    }
    __syncthreads();

    const char* nodes = reinterpret_cast<const char*>(lds);
    const float4* spheres = reinterpret_cast<const float4*>(lds + p.scene.off_spheres);
    // second centres + material bits: the tracers read a MOVING sphere's second centre from the LDS image (it is staged whenever a
    // sphere moves: xp.extra_in_lds); the shaders read the material bits from global memory / L1
    const float4* extra_lds = reinterpret_cast<const float4*>(lds + p.scene.off_extra);
    const float4* extra_g = reinterpret_cast<const float4*>(p.scene.blob + p.scene.off_extra);
    const float4* mats16 = reinterpret_cast<const float4*>(p.scene.blob + p.scene.off_mats);   // shader waves only: global memory / L1

#ifdef RT_PHASE_TIMERS
    unsigned long long pt_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pt_last_ = __builtin_readcyclecounter();
    uint32_t pc_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif

    if (is_tracer) {
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
        uint32_t idle_rounds = 0;
#ifdef RT_XCHG_GUARD
        uint32_t guard_steps = 0;
#endif

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
#ifdef RT_PHASE_TIMERS
                    pt_[5] += (unsigned long long)__popcll(__ballot(at_inner));   // lanes that took this step
                    pc_[5]++;
#endif
#ifdef RT_XCHG_GUARD
                    if (at_inner && ++guard_steps > 100000u) {   // debugging: a traversal that never ends = a corrupted ray
                        xc_st(ctrl + XC_ERR, 6u);
                        *xp.error_flag = 6u;
                        uint32_t* o = xp.error_flag + 16u + (blockIdx.x * (blockDim.x / 64u) + wave) * RT_XCHG_DEBUG_WORDS;
                        o[0] = 6u; o[1] = cur; o[2] = (uint32_t)(sp - stack) / 64u; o[3] = __float_as_uint(ray.d.x); o[4] = __float_as_uint(c_att.w); o[5] = depth; o[6] = lane; o[7] = __float_as_uint(ray.o.x);
                        cur = K_SHADE;
                        sp = stack + 64;
                    }
#endif
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
#ifdef RT_PHASE_TIMERS
                    pt_[6] += (unsigned long long)__popcll(m_leaf);
                    pc_[6]++;
#endif
                    if (at_leaf) {
                        const uint32_t code = cur & (K_LEAF - 1u);   // prim * 2 + is_moving
                        const uint32_t prim = code >> 1;
                        const float4 sph = spheres[prim];
                        f3 center = mk3(sph.x, sph.y, sph.z);
                        if (code & 1u) {
                            const float4 ex = extra_lds[prim];
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

#ifdef RT_PHASE_TIMERS
            pt_[7] += n_fin; pc_[7]++;   // finished lanes per exchange
#endif
            // ONE reservation for both directions: g_push finished traces -> SHADE ring, g_pop fresh rays <- TRACE ring
            const uint64_t m_emp = __ballot(cur == K_EMPTY);
            const uint32_t n_emp = (uint32_t)__popcll(m_emp);
            const uint32_t rank_fin = lane_rank(m_fin);
            uint32_t n_new = 0;
            bool do_push = false, do_pop = false;
            uint32_t ppos = 0, tpos = 0, pseq = 0, tseq = 0;
            {
                uint64_t w = xc_word(ring_word);
                for (uint32_t attempt = 0; attempt < 256u; attempt++) {
                    const uint32_t sq_h = XC_SQ_H(w), sq_t = XC_SQ_T(w), tq_h = XC_TQ_H(w), tq_t = XC_TQ_T(w);
                    const uint32_t g_push = min(n_fin, xp.sq_cap - ((sq_t - sq_h) & 0xffffu));
                    const uint32_t g_pop = min(n_emp + g_push, (tq_t - tq_h) & 0xffffu);   // the lanes that push are free for a new ray
                    do_push = false; do_pop = false;
                    if ((g_push | g_pop) == 0u) break;
                    do_push = cur == K_SHADE && rank_fin < g_push;
                    const bool free_after = cur == K_EMPTY || do_push;
                    const uint32_t rank_free = lane_rank(__ballot(free_after));
                    do_pop = free_after && rank_free < g_pop;
                    ppos = sq_t + rank_fin; tpos = tq_h + rank_free;
                    // sequence words of the positions this reservation grants IF it succeeds, loaded beside the compare-and-swap
                    if (do_push) pseq = xc_ld(sq_seq + (ppos & sq_mask));
                    if (do_pop) tseq = xc_ld(tq_seq + (tpos & tq_mask));
                    const uint64_t found = xc_cas(ring_word, w, xc_pack(sq_h, sq_t + g_push, tq_h + g_pop, tq_t), lane);
                    if (found == w) { n_new = g_pop; break; }
                    w = found;
                    do_push = false; do_pop = false;
                }
            }
            // entry writes of the pushing lanes and entry reads of the popping lanes go out together; ONE wait; then both slots' words
            uint4 a0 = make_uint4(0u, 0u, 0u, 0u), a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0;
            if (do_push) {
                const uint32_t slot = ppos & sq_mask;
                xc_wait_seq(sq_seq + slot, pseq, ppos, ctrl, xp.error_flag);
                XC_ORDER_ACQUIRE();
                sq_data[slot] = make_uint4(__float_as_uint(ray.o.x), __float_as_uint(ray.o.y), __float_as_uint(ray.o.z), __float_as_uint(ray.time));
                sq_data[xp.sq_cap + slot] = make_uint4(__float_as_uint(ray.d.x), __float_as_uint(ray.d.y), __float_as_uint(ray.d.z), __float_as_uint(rec_t));
                sq_data[2u * xp.sq_cap + slot] = make_uint4(__float_as_uint(c_att.x), __float_as_uint(c_att.y), __float_as_uint(c_att.z), __float_as_uint(c_att.w));
                sq_data[3u * xp.sq_cap + slot] = c_rng;
                sq_tail8[slot] = make_uint2((uint32_t)rec_code, depth);
            }
            if (do_pop) {
                const uint32_t slot = tpos & tq_mask;
                xc_wait_seq(tq_seq + slot, tseq, tpos + 1u, ctrl, xp.error_flag);
                XC_ORDER_ACQUIRE();
                a0 = tq_data[slot]; a1 = tq_data[xp.tq_cap + slot]; a2 = tq_data[2u * xp.tq_cap + slot];
                a3 = tq_data[3u * xp.tq_cap + slot]; a4 = tq_data[4u * xp.tq_cap + slot]; a5 = tq_data[5u * xp.tq_cap + slot];
            }
            XC_ORDER_RELEASE();   // every write has landed and every read has returned
            if (do_push) {
                xc_st(sq_seq + (ppos & sq_mask), ppos + 1u);
                cur = K_EMPTY;
            }
            if (do_pop) {
                xc_st(tq_seq + (tpos & tq_mask), tpos + xp.tq_cap);
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
#ifdef RT_XCHG_GUARD
                guard_steps = 0;
#endif
            }
            if (n_new != 0u && __ballot(!regular && cur < K_SHADE) != 0ull) irr_pending = true;
            XC_PT(3);
            if (__ballot(cur < K_SHADE) == 0ull && n_new == 0u) {   // nothing to trace and nothing came: the end, or the shaders are behind
                if (xc_ld(ctrl + XC_DONE) != 0u) break;
                if (xc_ld(ctrl + XC_ERR) != 0u || ++idle_rounds > RT_XCHG_IDLE_LIMIT) {
                    xc_st(ctrl + XC_ERR, 3u);
                    *xp.error_flag = 3u;
                    xc_dump(xp.error_flag, 3u, wave, lane, ctrl, (uint32_t)__popcll(__ballot(cur == K_SHADE)), (uint32_t)__popcll(__ballot(cur == K_EMPTY)), idle_rounds, 0u);
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
                XC_PT(4);
            } else {
                idle_rounds = 0;
            }
        }
    } else {
        // =====================================================================================================================
        // shader wave
        // =====================================================================================================================
        if (xp.shader_prio) __builtin_amdgcn_s_setprio(1);
        uint32_t pool_next = 0, pool_end = 0;
        bool pool_dry = false;   // this wave's pool is empty and the global sample counter is exhausted
        const f3 root_min = mk3(p.scene.root_min[0], p.scene.root_min[1], p.scene.root_min[2]);
        const f3 root_max = mk3(p.scene.root_max[0], p.scene.root_max[1], p.scene.root_max[2]);
        enum : uint32_t { S_NONE = 0, S_RAY = 1, S_NEED = 2 };

        for (;;) {
            // ---------------- (1) wait for work: finished traces to shade, or room in the population for new paths ---------------------
            uint32_t tries = 0;
            bool quit = false;
            uint64_t w = 0;
            for (;;) {
                w = xc_word(ring_word);
                const uint32_t avail = (XC_SQ_T(w) - XC_SQ_H(w)) & 0xffffu;
                if (avail >= xp.shade_min || (avail > 0u && tries >= xp.shade_patience)) break;
                const uint32_t pop = xc_ld(ctrl + XC_POP);
                if (avail == 0u && !pool_dry && pop < pop_target) break;          // start new paths
                if (xc_ld(ctrl + XC_DONE) != 0u) { quit = true; break; }
                if (xc_ld(ctrl + XC_ERR) != 0u || tries > RT_XCHG_IDLE_LIMIT) {
                    xc_st(ctrl + XC_ERR, 4u);
                    *xp.error_flag = 4u;
                    xc_dump(xp.error_flag, 4u, wave, lane, ctrl, avail, pop, tries, pool_dry ? 1u : 0u);
                    quit = true;
                    break;
                }
                if (avail == 0u && pop == 0u && xc_ld(ctrl + XC_DRY) == shard_shaders) {  // no ray alive in this shard and nobody can create one
                    xc_st(ctrl + XC_DONE, 1u);
                    quit = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
                tries++;
            }
            if (quit) break;
            XC_PT(8);

            uint32_t got = 0, spos = 0, sseq = 0;
            for (uint32_t attempt = 0; attempt < 256u; attempt++) {
                const uint32_t sq_h = XC_SQ_H(w);
                const uint32_t g = min(64u, (XC_SQ_T(w) - sq_h) & 0xffffu);
                if (g == 0u) break;
                spos = sq_h + lane;
                if (lane < g) sseq = xc_ld(sq_seq + (spos & sq_mask));
                const uint64_t found = xc_cas(ring_word, w, xc_pack(sq_h + g, XC_SQ_T(w), XC_TQ_H(w), XC_TQ_T(w)), lane);
                if (found == w) { got = g; break; }
                w = found;
            }
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
                const uint32_t slot = spos & sq_mask;
                xc_wait_seq(sq_seq + slot, sseq, spos + 1u, ctrl, xp.error_flag);
                XC_ORDER_ACQUIRE();
                const uint4 a0 = sq_data[slot], a1 = sq_data[xp.sq_cap + slot], a2 = sq_data[2u * xp.sq_cap + slot], a3 = sq_data[3u * xp.sq_cap + slot];
                const uint2 a4 = sq_tail8[slot];
                XC_ORDER_RELEASE();
                xc_st(sq_seq + slot, spos + xp.sq_cap);
                ray.o = mk3(__uint_as_float(a0.x), __uint_as_float(a0.y), __uint_as_float(a0.z)); ray.time = __uint_as_float(a0.w);
                ray.d = mk3(__uint_as_float(a1.x), __uint_as_float(a1.y), __uint_as_float(a1.z)); rec_t = __uint_as_float(a1.w);
                atten = mk3(__uint_as_float(a2.x), __uint_as_float(a2.y), __uint_as_float(a2.z)); out_idx = a2.w;
                rng.s0 = a3.x; rng.s1 = a3.y; rng.s2 = a3.z; rng.s3 = a3.w;
                rec_code = (int32_t)a4.x; depth = a4.y;
            }
            XC_PT(9);
#ifdef RT_PHASE_TIMERS
            pt_[14] += got; pc_[14]++;   // finished traces per shade round
#endif

            // ---------------- (2) sample_world's loop body after the trace (Renderer.cu:149-176) ------------------------------------
            if (lane < got) {
                if (rec_code < 0) {
                    const float t = normalize(ray.d).y * 0.5f + 0.5f;
                    const f3 sky = linear_interpolate(mk3(0.1f, 0.2f, 0.4f), mk3(0.9f, 0.9f, 0.99f), t);
                    const f3 rad = atten * sky;
                    store_sample(p.samples, out_idx, rad.x, rad.y, rad.z);
                    state = S_NEED;
                } else if (depth + 1u >= p.max_depth) {
                    store_sample(p.samples, out_idx, 0.0f, 0.0f, 0.0f);   // the scatter of the last allowed bounce cannot reach the sky
                    state = S_NEED;
                } else {
                    const f3 hit_p = ray_at(ray, rec_t);
                    const uint32_t prim = (uint32_t)rec_code >> 1;
                    const float4 sph = spheres[prim];
                    const float4 ex = extra_g[prim];
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
                        store_sample(p.samples, out_idx, 0.0f, 0.0f, 0.0f);
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
                    if (xc_ld(ctrl + XC_POP) < pop_target) {
                        const uint32_t old = xc_add(ctrl + XC_POP, n_free);
                        allowed = old >= pop_target ? 0u : min(n_free, pop_target - old);
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
                    const uint4 rs = RT_LOAD_ONCE(p.prim_rng + n);
                    const float4 po = RT_LOAD_ONCE(p.prim_o + n), pd = RT_LOAD_ONCE(p.prim_d + n);   // together: one global round trip
                    if ((rs.x | rs.y | rs.z | rs.w) != 0u) {
                        out_idx = n;
                        ray.o = mk3(po.x, po.y, po.z); ray.d = mk3(pd.x, pd.y, pd.z); ray.time = po.w;
                        rng.s0 = rs.x; rng.s1 = rs.y; rng.s2 = rs.z; rng.s3 = rs.w;
                        atten = mk3(1.0f);
                        depth = 0;
                        if (p.max_depth == 0u) store_sample(p.samples, n, 0.0f, 0.0f, 0.0f);   // and the lane asks again
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
                uint32_t pushed = 0, spins = 0, cas_lost = 0;
                uint64_t pw = n_ray != 0u ? xc_word(ring_word) : 0ull;
                while (pushed < n_ray) {
                    const uint32_t tq_h = XC_TQ_H(pw), tq_t = XC_TQ_T(pw);
                    const uint32_t g = min(n_ray - pushed, xp.tq_cap - ((tq_t - tq_h) & 0xffffu));
                    if (g == 0u) {
                        __builtin_amdgcn_s_sleep(2);
                        if (xc_ld(ctrl + XC_ERR) != 0u || ++spins > RT_XCHG_IDLE_LIMIT) {
                            xc_st(ctrl + XC_ERR, 5u);
                            *xp.error_flag = 5u;
                            xc_dump(xp.error_flag, 5u, wave, lane, ctrl, n_ray, pushed, spins, 0u);
                            break;
                        }
                        pw = xc_word(ring_word);
                        continue;
                    }
                    const bool mine = state == S_RAY && rank >= pushed && rank < pushed + g;
                    const uint32_t pos = tq_t + (rank - pushed), slot = pos & tq_mask;
                    uint32_t have = 0;
                    if (mine) have = xc_ld(tq_seq + slot);
                    const uint64_t found = xc_cas(ring_word, pw, xc_pack(XC_SQ_H(pw), XC_SQ_T(pw), tq_h, tq_t + g), lane);
                    if (found != pw) {
                        // a lost compare-and-swap is retried with the word it found — a bounded number of times in a row like every other wait of
                        // this kernel (a corrupted `found` once made exactly this loop spin for ever: the readfirstlane sign extension, EXPERIMENTS.md)
                        if (++cas_lost > RT_XCHG_IDLE_LIMIT || xc_ld(ctrl + XC_ERR) != 0u) {
                            xc_st(ctrl + XC_ERR, 5u);
                            *xp.error_flag = 5u;
                            xc_dump(xp.error_flag, 5u, wave, lane, ctrl, n_ray, pushed, cas_lost, 1u);
                            break;
                        }
                        pw = found;
                        continue;
                    }
                    cas_lost = 0;
                    if (mine) {
                        xc_wait_seq(tq_seq + slot, have, pos, ctrl, xp.error_flag);
                        XC_ORDER_ACQUIRE();
                        tq_data[slot] = make_uint4(__float_as_uint(ray.o.x), __float_as_uint(ray.o.y), __float_as_uint(ray.o.z), __float_as_uint(ray.time));
                        tq_data[xp.tq_cap + slot] = make_uint4(__float_as_uint(ray.d.x), __float_as_uint(ray.d.y), __float_as_uint(ray.d.z), __float_as_uint(ray_a));
                        tq_data[2u * xp.tq_cap + slot] = make_uint4(__float_as_uint(inv_d.x), __float_as_uint(inv_d.y), __float_as_uint(inv_d.z), word);
                        tq_data[3u * xp.tq_cap + slot] = make_uint4(__float_as_uint(inv_lo.x), __float_as_uint(inv_lo.y), __float_as_uint(inv_lo.z), depth);
                        tq_data[4u * xp.tq_cap + slot] = make_uint4(__float_as_uint(atten.x), __float_as_uint(atten.y), __float_as_uint(atten.z), out_idx);
                        tq_data[5u * xp.tq_cap + slot] = make_uint4(rng.s0, rng.s1, rng.s2, rng.s3);
                        XC_ORDER_RELEASE();
                        xc_st(tq_seq + slot, pos + 1u);
                    }
                    pushed += g;
                    pw = xc_pack(XC_SQ_H(pw), XC_SQ_T(pw), tq_h, tq_t + g);   // what the word was just set to: the next round starts from it
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

// rt_device.hip — the device half of the C ABI: Renderer (kernel selection, passes, launches, per-kernel timers), shard layout, assembly.
// Probes and self-tests: rt_probes.hip.  Multi-GPU driver: rt_multi.hip.
#include "rt_runtime.hpp"
#include "rt_render_kernels.hpp"
#include "rt_stream_kernel.hpp"
#include "rt_xchg_kernel.hpp"

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
    bool tol = false;            // variant 3 with the tolerance-mode box test (requested as variant 6)
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
        if (want > 6) return rt_fail(RT_ERR_INVALID, "unknown kernel variant %u", want);
        const bool want_tol = want == 6;   // variant 6: variant 3 with the tolerance-mode box test (opt-in; inside |delta| < 1e-3, not bit-exact by construction)
        if (want_tol) want = 3;
        bool can_stream = scene.has_packed;
        if (can_stream) {
            stream_block = RT_STREAM_BLOCK;
            if (scene.big) {  // the per-lane stacks (32-bit entries) and, in what two workgroups per CU leave free, the top of the tree
                stream_block = RT_STREAM_BLOCK;
                const uint32_t stacks = (stream_block * scene.packed.stack_cap * (scene.wide ? 4u : 2u) + 63u) & ~63u;
                const uint32_t budget = stacks + 4096u <= lds_per_cu / 2u ? lds_per_cu / 2u : lds_per_cu;
                uint32_t top_bytes = budget > stacks ? budget - stacks : 0u;
                top_bytes = std::min(top_bytes & ~63u, scene.packed.n_inner * (RT_NODE_DWORDS_BIG * 4u));
                if (const char* env = std::getenv("RT06_TOP_NODES")) top_bytes = std::min(top_bytes, (uint32_t)std::atoi(env) * (RT_NODE_DWORDS_BIG * 4u));
                if (scene.queue) top_bytes = 0;   // the queue walk reads the flat world's own nodes
                n_top = top_bytes / (RT_NODE_DWORDS_BIG * 4u);
                stream_lds_bytes = top_bytes + stacks;
            } else {
                stream_lds_bytes = scene.packed.blob_vec4 * 16u + stream_block * scene.packed.stack_cap * 2u;
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
        if (want_tol) {
            // Kept for the reference's own feature set only (spheres: a sphere touches its box at six points, so a box decision that flips by an ulp
            // almost never meets a hit).  Worlds with quads are refused: a quad's edges ARE its box's edges, and the Cornell box at its own 5000 spp moved
            // one pixel by 2.1e-3, outside the tolerance; the global-memory form gains 8 %, below the 15 % it would have to (EXPERIMENTS.md E4)
            if (scene.big || scene.extended)
                return rt_fail(RT_ERR_INVALID, "kernel variant 6 (tolerance-mode box test) is instantiated for LDS-resident RT_WORLD_BVH worlds of the reference's feature set only (spheres, the three scattering materials, the sky): use variant 0");
            tol = true;
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
        if (tol) return reinterpret_cast<const void*>(&render_kernel_stream<false, false, 768, RT_WORLD_BVH, 0, false, false, true>);
        if (variant == 2) return reinterpret_cast<const void*>(&render_kernel_stream<true, false, 768>);
        if (variant == 4) return reinterpret_cast<const void*>(&render_kernel_stream<false, true, 768>);
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
            // it keeps the next frame's generator from moving in beside the persistent kernel's main phase (measured harmful, EXPERIMENTS.md E2) and
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
    out[0] = r->tol ? 6u : r->variant;
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


hipStream_t rt_renderer_own_stream(rt_renderer* r) { return r->stream; }
float* rt_renderer_own_framebuffer(rt_renderer* r) { return r->fb.as<float>(); }
int rt_renderer_check_device_flags(rt_renderer* r) {
    const int rc = check_xchg_error(r->xchg_error);
    return rc != RT_OK ? rc : check_traversal_overflow(r->scene);
}

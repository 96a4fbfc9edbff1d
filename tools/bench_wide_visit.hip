// Microbenchmark for SURVEY §8f rank 4 ("wider BVH4/8 nodes"): what does ONE 4-wide node visit cost on gfx950 next to the two
// binary visits it replaces, at the streaming kernel's occupancy (768-thread workgroups, 2 per CU = 6 waves per SIMD), with the
// nodes in the LDS and the production arithmetic (exact quotients from a two-word reciprocal, rt_fastdiv.hpp)?
//
// Both kernels walk the SAME geometry: a median-split binary tree over 256 random boxes (depth 8; the Book scenes: 488 leaves, depth 10), stored (a) as the
// production 19-dword wide nodes — both child boxes as (min, max, min) triples + two 16-bit references — and (b) collapsed two
// levels at a time into 4-wide nodes: four grandchild boxes as triples (36 dwords) + four references (2 dwords).  The binary
// kernel's visit is the production hot-loop body (fetch_wide_node + slab_near_far_regular + near-first push, BVH.cu:87-96); the
// 4-wide visit is the same rule generalised (all four child boxes, nearest first, the others pushed far-to-near, culling at push
// time only).  Leaves only count (no primitive test), so both kernels reach the same set of leaves — checked — and differ only in
// the visits.  Reported: kernel time, wave-steps, lane-visits, SIMD issue cycles per wave-step (time x clock x SIMDs / steps:
// the SIMDs are saturated at this occupancy) and the whole-traversal ratio.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -Iinclude -Iray-tracing-v06_amd/csrc -o build/bench_wide_visit tools/bench_wide_visit.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "rt_stream_kernel.hpp"

int rt_fail(int code, const char*, ...) { return code; }   // rt_internal.hpp's hook (unused here)

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr uint32_t LEAF = 0x8000u, SENTINEL = 0xffffu;
constexpr uint32_t W4_DWORDS = 38u;   // 4 boxes x 3 axes x (min, max, min) + 2 dwords of references
constexpr int BLOCK = 768;

struct Params {
    const uint4* blob;      // LDS image: nodes
    uint32_t blob_vec4;
    uint32_t root;
    uint32_t stack_cap;
    uint32_t rays_per_lane;
    const float4* ray_o;    // per (thread, ray)
    const float4* ray_d;
    unsigned long long* out;   // [0] wave-steps, [1] lane-visits, [2] leaf checksum, [3] leaves
};

__device__ __forceinline__ void begin_ray(const Params& p, uint32_t gid, uint32_t k, Ray& ray, f3& inv_d, f3& inv_lo, uint32_t& kx, uint32_t& ky, uint32_t& kz) {
    const float4 o = p.ray_o[(size_t)k * gridDim.x * BLOCK + gid], d = p.ray_d[(size_t)k * gridDim.x * BLOCK + gid];
    ray.o = mk3(o.x, o.y, o.z); ray.d = mk3(d.x, d.y, d.z); ray.time = 0.0f;
    inv_d = mk3(rcp_exact_regular(ray.d.x), rcp_exact_regular(ray.d.y), rcp_exact_regular(ray.d.z));
    inv_lo = mk3(rcp_low_word(ray.d.x, inv_d.x), rcp_low_word(ray.d.y, inv_d.y), rcp_low_word(ray.d.z, inv_d.z));
    kx = (__float_as_uint(ray.d.x) >> 29) & 4u; ky = (__float_as_uint(ray.d.y) >> 29) & 4u; kz = (__float_as_uint(ray.d.z) >> 29) & 4u;
}

// ---- binary: the production visit ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK, BLOCK / 128) void walk_binary(Params p) {
    extern __shared__ uint4 lds[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, gid = blockIdx.x * BLOCK + tid;
    for (uint32_t i = tid; i < p.blob_vec4; i += BLOCK) lds[i] = p.blob[i];
    __syncthreads();
    const char* nodes = reinterpret_cast<const char*>(lds);
    uint16_t* stack = reinterpret_cast<uint16_t*>(lds + p.blob_vec4) + wave * 64u * p.stack_cap + lane;
    *stack = (uint16_t)SENTINEL;
    unsigned long long steps = 0, visits = 0, sum = 0, leaves = 0;
    Ray ray; f3 inv_d, inv_lo; uint32_t kx, ky, kz;
    uint32_t k = 0;
    begin_ray(p, gid, k, ray, inv_d, inv_lo, kx, ky, kz);
    uint32_t cur = p.root;
    uint16_t* sp = stack + 64;
    const float rec_t = RT_MISS_DIST;
    for (;;) {
        const bool at_inner = cur < LEAF;
        const uint64_t m = __ballot(at_inner);
        if (m != 0ull) {
            steps++;
            visits += (unsigned long long)__popcll(m);
        }
        if (at_inner) {
            const WideNodeData nd = fetch_wide_node<false>(nodes, lds, 0u, cur, kx, ky, kz);
            float tl, tr;
            const bool hl = slab_near_far_regular(nd.lnx, nd.lny, nd.lnz, nd.lfx, nd.lfy, nd.lfz, ray, inv_d, inv_lo, rec_t, tl);
            const bool hr = slab_near_far_regular(nd.rnx, nd.rny, nd.rnz, nd.rfx, nd.rfy, nd.rfz, ray, inv_d, inv_lo, rec_t, tr);
            const bool go_right = hr && (!hl || tl > tr);
            if (hl && hr) { *sp = (uint16_t)(go_right ? nd.left : nd.right); sp += 64; }
            cur = go_right ? nd.right : nd.left;
            if (!(hl || hr)) { sp -= 64; cur = *sp; }
        } else if (cur != SENTINEL) {   // a leaf: count it, pop
            sum += cur & 0x7fffu; leaves++;
            sp -= 64; cur = *sp;
        } else if (k + 1 < p.rays_per_lane) {   // next ray of this lane
            k++;
            begin_ray(p, gid, k, ray, inv_d, inv_lo, kx, ky, kz);
            cur = p.root; sp = stack + 64;
        }
        if (__ballot(cur != SENTINEL || k + 1 < p.rays_per_lane) == 0ull) break;
    }
    if (lane == 0) { atomicAdd(p.out + 0, steps); }
    atomicAdd(p.out + 2, sum); atomicAdd(p.out + 3, leaves);
    if (lane == 0) atomicAdd(p.out + 1, visits);
}

// ---- 4-wide ----------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool slab4(const float* q, uint32_t kxyz_off_x, uint32_t kxyz_off_y, uint32_t kxyz_off_z, const Ray& ray, f3 inv_d, f3 inv_lo, float maxd, float& tmin) {
    // q: the 9 dwords of one child box; (near, far) of an axis are two consecutive dwords at offset 0 or 1 of its triple
    const float nx = q[kxyz_off_x], fx = q[kxyz_off_x + 1], ny = q[3 + kxyz_off_y], fy = q[3 + kxyz_off_y + 1], nz = q[6 + kxyz_off_z], fz = q[6 + kxyz_off_z + 1];
    return slab_near_far_regular(nx, ny, nz, fx, fy, fz, ray, inv_d, inv_lo, maxd, tmin);
}

__global__ __launch_bounds__(BLOCK, BLOCK / 128) void walk_wide4(Params p) {
    extern __shared__ uint4 lds[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, gid = blockIdx.x * BLOCK + tid;
    for (uint32_t i = tid; i < p.blob_vec4; i += BLOCK) lds[i] = p.blob[i];
    __syncthreads();
    const float* nodes = reinterpret_cast<const float*>(lds);
    uint16_t* stack = reinterpret_cast<uint16_t*>(lds + p.blob_vec4) + wave * 64u * p.stack_cap + lane;
    *stack = (uint16_t)SENTINEL;
    unsigned long long steps = 0, visits = 0, sum = 0, leaves = 0;
    Ray ray; f3 inv_d, inv_lo; uint32_t kx, ky, kz;
    uint32_t k = 0;
    begin_ray(p, gid, k, ray, inv_d, inv_lo, kx, ky, kz);
    uint32_t cur = p.root;
    uint16_t* sp = stack + 64;
    const float rec_t = RT_MISS_DIST;
    for (;;) {
        const bool at_inner = cur < LEAF;
        const uint64_t m = __ballot(at_inner);
        if (m != 0ull) {
            steps++;
            visits += (unsigned long long)__popcll(m);
        }
        if (at_inner) {
            const float* q = nodes + cur * W4_DWORDS;
            const uint32_t ox = kx >> 2, oy = ky >> 2, oz = kz >> 2;
            float t0, t1, t2, t3;
            const bool h0 = slab4(q, ox, oy, oz, ray, inv_d, inv_lo, rec_t, t0);
            const bool h1 = slab4(q + 9, ox, oy, oz, ray, inv_d, inv_lo, rec_t, t1);
            const bool h2 = slab4(q + 18, ox, oy, oz, ray, inv_d, inv_lo, rec_t, t2);
            const bool h3 = slab4(q + 27, ox, oy, oz, ray, inv_d, inv_lo, rec_t, t3);
            const uint32_t r01 = reinterpret_cast<const uint32_t*>(q)[36], r23 = reinterpret_cast<const uint32_t*>(q)[37];
            // sort the four (distance, reference) pairs, missed children last (distance = _MISS_DIST): a 5-comparator network
            float d0 = h0 ? t0 : RT_MISS_DIST, d1 = h1 ? t1 : RT_MISS_DIST, d2 = h2 ? t2 : RT_MISS_DIST, d3 = h3 ? t3 : RT_MISS_DIST;
            uint32_t c0 = r01 & 0xffffu, c1 = r01 >> 16, c2 = r23 & 0xffffu, c3 = r23 >> 16;
#define CSWAP(da, ca, db, cb) do { const bool s_ = da > db; const float td_ = s_ ? db : da; db = s_ ? da : db; da = td_; const uint32_t tc_ = s_ ? cb : ca; cb = s_ ? ca : cb; ca = tc_; } while (0)
            CSWAP(d0, c0, d1, c1); CSWAP(d2, c2, d3, c3); CSWAP(d0, c0, d2, c2); CSWAP(d1, c1, d3, c3); CSWAP(d1, c1, d2, c2);
#undef CSWAP
            const uint32_t n_hit = (h0 ? 1u : 0u) + (h1 ? 1u : 0u) + (h2 ? 1u : 0u) + (h3 ? 1u : 0u);
            // push far-to-near what is not visited next (culling at push time only: a hit child has dist < rec.distance)
            if (n_hit > 3u) { *sp = (uint16_t)c3; sp += 64; }
            if (n_hit > 2u) { *sp = (uint16_t)c2; sp += 64; }
            if (n_hit > 1u) { *sp = (uint16_t)c1; sp += 64; }
            cur = c0;
            if (n_hit == 0u) { sp -= 64; cur = *sp; }
        } else if (cur != SENTINEL) {
            sum += cur & 0x7fffu; leaves++;
            sp -= 64; cur = *sp;
        } else if (k + 1 < p.rays_per_lane) {
            k++;
            begin_ray(p, gid, k, ray, inv_d, inv_lo, kx, ky, kz);
            cur = p.root; sp = stack + 64;
        }
        if (__ballot(cur != SENTINEL || k + 1 < p.rays_per_lane) == 0ull) break;
    }
    if (lane == 0) { atomicAdd(p.out + 0, steps); }
    atomicAdd(p.out + 2, sum); atomicAdd(p.out + 3, leaves);
    if (lane == 0) atomicAdd(p.out + 1, visits);
}

// ---- host: the tree ------------------------------------------------------------------------------------------------------------
struct Box { float mn[3], mx[3]; };
struct BNode { Box b; int left, right, leaf; };
static std::vector<BNode> g_nodes;
static int build(std::vector<int>& idx, int lo, int hi, const std::vector<Box>& boxes) {
    BNode n{};
    for (int a = 0; a < 3; a++) { n.b.mn[a] = 1e30f; n.b.mx[a] = -1e30f; }
    for (int i = lo; i < hi; i++) for (int a = 0; a < 3; a++) { n.b.mn[a] = std::min(n.b.mn[a], boxes[idx[i]].mn[a]); n.b.mx[a] = std::max(n.b.mx[a], boxes[idx[i]].mx[a]); }
    n.left = n.right = -1; n.leaf = -1;
    if (hi - lo == 1) { n.leaf = idx[lo]; g_nodes.push_back(n); return (int)g_nodes.size() - 1; }
    int axis = 0;
    for (int a = 1; a < 3; a++) if (n.b.mx[a] - n.b.mn[a] > n.b.mx[axis] - n.b.mn[axis]) axis = a;
    std::sort(idx.begin() + lo, idx.begin() + hi, [&](int x, int y) { return boxes[x].mn[axis] < boxes[y].mn[axis]; });
    const int mid = (lo + hi) / 2;
    const int l = build(idx, lo, mid, boxes), r = build(idx, mid, hi, boxes);
    n.left = l; n.right = r;
    g_nodes.push_back(n);
    return (int)g_nodes.size() - 1;
}
static void put_box(uint32_t* d, const Box& b) {
    for (int a = 0; a < 3; a++) { memcpy(d + 3 * a, &b.mn[a], 4); memcpy(d + 3 * a + 1, &b.mx[a], 4); memcpy(d + 3 * a + 2, &b.mn[a], 4); }
}

int main(int argc, char** argv) {
    const int n_leaves = 256, rays_per_lane = argc > 1 ? atoi(argv[1]) : 48;
    std::mt19937 rng(1984);
    std::uniform_real_distribution<float> U(0.0f, 1.0f);
    std::vector<Box> boxes(n_leaves);
    for (auto& b : boxes)
        for (int a = 0; a < 3; a++) { const float c = U(rng) * 20.0f - 10.0f, h = 0.4f + U(rng) * 0.8f; b.mn[a] = c - h; b.mx[a] = c + h; }
    std::vector<int> idx(n_leaves);
    for (int i = 0; i < n_leaves; i++) idx[i] = i;
    const int root = build(idx, 0, n_leaves, boxes);
    // (a) binary wide nodes
    std::vector<int> wide_of(g_nodes.size(), -1);
    int n_inner = 0;
    for (size_t i = 0; i < g_nodes.size(); i++) if (g_nodes[i].leaf < 0) wide_of[i] = n_inner++;
    auto ref2 = [&](int node) -> uint32_t { return g_nodes[node].leaf >= 0 ? (LEAF | (uint32_t)g_nodes[node].leaf) : (uint32_t)wide_of[node]; };
    std::vector<uint32_t> bin((size_t)((n_inner * RT_NODE_DWORDS + 3) / 4) * 4, 0u);
    for (size_t i = 0; i < g_nodes.size(); i++) {
        if (wide_of[i] < 0) continue;
        uint32_t* d = bin.data() + (size_t)wide_of[i] * RT_NODE_DWORDS;
        put_box(d, g_nodes[g_nodes[i].left].b); put_box(d + 9, g_nodes[g_nodes[i].right].b);
        d[RT_NODE_REFS] = ref2(g_nodes[i].left) | (ref2(g_nodes[i].right) << 16);
    }
    // (b) 4-wide nodes: a binary inner node at an even level + its two children (n_leaves is a power of four: every grandchild exists)
    std::vector<int> w4_of(g_nodes.size(), -1);
    int n_w4 = 0;
    std::vector<std::pair<int, int>> todo{{root, 0}};
    std::vector<int> order;
    while (!todo.empty()) {
        auto [ni, lvl] = todo.back(); todo.pop_back();
        if (g_nodes[ni].leaf >= 0) continue;
        if (lvl % 2 == 0) { w4_of[ni] = n_w4++; order.push_back(ni); }
        todo.push_back({g_nodes[ni].left, lvl + 1}); todo.push_back({g_nodes[ni].right, lvl + 1});
    }
    auto ref4 = [&](int node) -> uint32_t { return g_nodes[node].leaf >= 0 ? (LEAF | (uint32_t)g_nodes[node].leaf) : (uint32_t)w4_of[node]; };
    std::vector<uint32_t> w4((size_t)((n_w4 * W4_DWORDS + 3) / 4) * 4, 0u);
    for (int ni : order) {
        uint32_t* d = w4.data() + (size_t)w4_of[ni] * W4_DWORDS;
        const int l = g_nodes[ni].left, r = g_nodes[ni].right;
        const int gc[4] = {g_nodes[l].left, g_nodes[l].right, g_nodes[r].left, g_nodes[r].right};
        for (int c = 0; c < 4; c++) put_box(d + 9 * c, g_nodes[gc[c]].b);
        d[36] = ref4(gc[0]) | (ref4(gc[1]) << 16); d[37] = ref4(gc[2]) | (ref4(gc[3]) << 16);
    }
    printf("tree: %d leaves, %d binary wide nodes (%zu KB), %d 4-wide nodes (%zu KB)\n", n_leaves, n_inner, bin.size() * 4 / 1024, n_w4, w4.size() * 4 / 1024);

    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount * 2;
    const size_t n_threads = (size_t)grid * BLOCK, n_rays = n_threads * rays_per_lane;
    std::vector<float4> ro(n_rays), rd(n_rays);
    for (size_t i = 0; i < n_rays; i++) {
        float o[3], t[3];
        for (int a = 0; a < 3; a++) { o[a] = (U(rng) * 2.0f - 1.0f) * 16.0f; t[a] = (U(rng) * 2.0f - 1.0f) * 9.0f; }
        ro[i] = make_float4(o[0], o[1], o[2], 0.0f);
        float d[3];
        for (int a = 0; a < 3; a++) { d[a] = t[a] - o[a]; if (fabsf(d[a]) < 1e-3f) d[a] = 1e-3f; }
        rd[i] = make_float4(d[0], d[1], d[2], 0.0f);
    }
    float4 *d_ro, *d_rd; uint4 *d_bin, *d_w4; unsigned long long* d_out;
    CHECK(hipMalloc(&d_ro, n_rays * 16)); CHECK(hipMalloc(&d_rd, n_rays * 16));
    CHECK(hipMemcpy(d_ro, ro.data(), n_rays * 16, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_rd, rd.data(), n_rays * 16, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_bin, bin.size() * 4)); CHECK(hipMemcpy(d_bin, bin.data(), bin.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_w4, w4.size() * 4)); CHECK(hipMemcpy(d_w4, w4.data(), w4.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_out, 64));
    const double clock_ghz = prop.clockRate / 1e6, n_simd = prop.multiProcessorCount * 4.0;
    double ms_of[2] = {0, 0}, steps_of[2] = {0, 0};
    unsigned long long leaves_of[2] = {0, 0}, sum_of[2] = {0, 0};
    for (int which = 0; which < 2; which++) {
        Params p;
        p.blob = which == 0 ? d_bin : d_w4;
        p.blob_vec4 = (uint32_t)((which == 0 ? bin.size() : w4.size()) / 4);
        p.root = which == 0 ? (uint32_t)wide_of[root] : (uint32_t)w4_of[root];
        p.stack_cap = which == 0 ? 10u : 14u;   // depth 8 binary: <= 8 pending + sentinel; 4 levels x 3 pending + sentinel
        p.rays_per_lane = (uint32_t)rays_per_lane;
        p.ray_o = d_ro; p.ray_d = d_rd; p.out = d_out;
        const size_t lds_bytes = (size_t)p.blob_vec4 * 16 + (size_t)BLOCK * p.stack_cap * 2;
        if (lds_bytes > 80 * 1024) { fprintf(stderr, "LDS image of %zu B does not leave two workgroups per CU\n", lds_bytes); return 1; }
        const void* fn = which == 0 ? (const void*)walk_binary : (const void*)walk_wide4;
        CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        float best = 1e30f;
        unsigned long long h[4] = {0, 0, 0, 0};
        for (int rep = 0; rep < 4; rep++) {
            CHECK(hipMemset(d_out, 0, 64));
            CHECK(hipEventRecord(e0));
            void* args[] = {&p};
            CHECK(hipLaunchKernel(fn, dim3(grid), dim3(BLOCK), args, lds_bytes, 0));
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
            CHECK(hipMemcpy(h, d_out, 32, hipMemcpyDeviceToHost));
        }
        ms_of[which] = best; steps_of[which] = (double)h[0]; leaves_of[which] = h[3]; sum_of[which] = h[2];
        const double simd_cycles = best * 1e-3 * clock_ghz * 1e9 * n_simd;
        printf("%-8s LDS %5zu B/workgroup  %8.3f ms  wave-steps %.4g  lane-visits %.4g (%.1f lanes per step)  leaves reached %llu\n"
               "         SIMD issue cycles per wave-step %.1f (at %.2f GHz x %d SIMDs)  lane-visits per ray %.2f\n",
               which == 0 ? "binary" : "4-wide", lds_bytes, best, (double)h[0], (double)h[1], (double)h[1] / (double)h[0], h[3],
               simd_cycles / (double)h[0], clock_ghz, (int)n_simd, (double)h[1] / (double)n_rays);
    }
    printf("same leaves reached: %s (checksum %s)\n", leaves_of[0] == leaves_of[1] ? "yes" : "NO", sum_of[0] == sum_of[1] ? "equal" : "DIFFERENT");
    printf("whole traversal, 4-wide / binary: time %.3f, wave-steps %.3f; one 4-wide step costs %.2f binary steps\n", ms_of[1] / ms_of[0],
           steps_of[1] / steps_of[0], (ms_of[1] / steps_of[1]) / (ms_of[0] / steps_of[0]));
    return 0;
}

// rt_render_kernels.hpp — the render kernels (twin of render_kernel, main/src/Renderer.cu:183-217).
//
// The reference runs one thread per PIXEL with a serial spp loop and keeps a 48-byte XORWOW state per
// pixel in global memory.  Here one work-item owns one pixel-SAMPLE: the 64 lanes of a wavefront work
// on samples of the same pixel (coherent primary rays), the counter-based RNG needs no state, and the
// per-pixel mean is a cross-lane reduction.
#pragma once
#include <hip/hip_runtime.h>

#include "rt06.h"
#include "rt_device_funcs.hpp"
#include "rt_internal.hpp"
#include "rt_layout.hpp"

struct RenderParams {
    uint32_t width, height, spp, max_depth;
    uint64_t seed;
    rt_camera cam;
    DeviceWorld world;
    TileMap tm;
    float* out;
    uint32_t* work_counter;
};

__device__ __forceinline__ bool local_pixel_to_gid(const TileMap& tm, uint32_t L, uint32_t& gid) {
    uint32_t tl = L / (RT_TILE * RT_TILE), p = L % (RT_TILE * RT_TILE);
    uint32_t gt = tl * tm.world_size + tm.rank;
    if (gt >= tm.n_tiles) return false;
    uint32_t x = (gt % tm.tiles_x) * RT_TILE + (p % RT_TILE);
    uint32_t y = (gt / tm.tiles_x) * RT_TILE + (p / RT_TILE);
    if (x >= tm.width || y >= tm.height) return false;
    gid = y * tm.width + x;
    return true;
}

// mean, clamp, sqrt-gamma, alpha = 1 (Renderer.cu:206-216)
__device__ __forceinline__ void write_pixel(const RenderParams& p, uint32_t L, uint32_t gid, f3 radiance_sum) {
    f3 radiance = radiance_sum * (1.0f / (float)p.spp);
    f3 col = clamp01_sqrt(radiance);
    float4 o = make_float4(col.x, col.y, col.z, 1.0f);
    reinterpret_cast<float4*>(p.out)[p.tm.direct ? gid : L] = o;
}

__device__ __forceinline__ f3 wave_sum(f3 a) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        a.x += __shfl_xor(a.x, off);
        a.y += __shfl_xor(a.y, off);
        a.z += __shfl_xor(a.z, off);
    }
    return a;
}

// ---------------------------------------------------------------------------------------------
// variant 1 — baseline: one wavefront per pixel, lane l takes samples l, l+64, ...; world read from
// global memory (L1/L2-resident: 31 KB of nodes + 16 KB of primitives); traversal stack in scratch.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void render_kernel_wave_per_pixel(RenderParams p) {
    uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t lane = threadIdx.x & 63u;
    if (wave >= p.tm.n_local_tiles * RT_TILE * RT_TILE) return;
    uint32_t gid;
    if (!local_pixel_to_gid(p.tm, wave, gid)) return;
    f3 acc = mk3(0.0f);
    for (uint32_t s = lane; s < p.spp; s += 64u)
        acc = acc + one_sample(p.world, p.cam, p.width, p.height, p.max_depth, p.seed, gid, s);
    acc = wave_sum(acc);
    if (lane == 0) write_pixel(p, wave, gid, acc);
}

static int launch_render(const RenderParams& p, uint32_t variant, hipStream_t st) {
    uint32_t n_local_pixels = p.tm.n_local_tiles * RT_TILE * RT_TILE;
    (void)variant;
    uint32_t blocks = (n_local_pixels + 3) / 4;
    render_kernel_wave_per_pixel<<<blocks, 256, 0, st>>>(p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rt_fail(RT_ERR_HIP, "render kernel launch failed: %s", hipGetErrorString(e));
    return RT_OK;
}

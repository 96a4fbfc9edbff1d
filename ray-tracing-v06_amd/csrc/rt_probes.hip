// rt_probes.hip — rt_probe_* (one hot-path function per call over arrays: the parity tests' entry points) and rt_selftest_* (exhaustive checks of
// the exact-division building blocks).  Test entry points of the C ABI; nothing here is on the render path.
#include "rt_runtime.hpp"
#include "rt_fastdiv.hpp"

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

// The hot loop's box pair (rt_fastdiv.hpp: CERTIFIED FAR PLANES) next to the verbatim box tests: near parameters as exact quotients, far parameters as
// products whose `tmin <= tmax` decisions are certified — a lane that cannot certify redoes its far planes exactly (the kernel does that for the whole wave).
__global__ void probe_boxpair_certified_kernel(size_t n, const float* boxes, const float* rays, const float* maxd, int32_t* out) {
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
    const f3 inv_d = mk3(rcp_exact_regular(r.d.x), rcp_exact_regular(r.d.y), rcp_exact_regular(r.d.z));
    const f3 inv_lo = mk3(rcp_low_word(r.d.x, inv_d.x), rcp_low_word(r.d.y, inv_d.y), rcp_low_word(r.d.z, inv_d.z));
    // near = the plane the ray enters through (box min where d >= 0), as the kernel's (min, max, min) triples deliver it
    const bool sx = r.d.x < 0, sy = r.d.y < 0, sz = r.d.z < 0;
    const float lnx = sx ? b[3] : b[0], lny = sy ? b[4] : b[1], lnz = sz ? b[5] : b[2], lfx = sx ? b[0] : b[3], lfy = sy ? b[1] : b[4], lfz = sz ? b[2] : b[5];
    const float rnx = sx ? b[9] : b[6], rny = sy ? b[10] : b[7], rnz = sz ? b[11] : b[8], rfx = sx ? b[6] : b[9], rfy = sy ? b[7] : b[10], rfz = sz ? b[8] : b[11];
    const float tl = slab_near_exact(lnx, lny, lnz, r, inv_d, inv_lo), tr = slab_near_exact(rnx, rny, rnz, r, inv_d, inv_lo);
    float far_l = slab_far_product(lfx, lfy, lfz, r, inv_d), far_r = slab_far_product(rfx, rfy, rfz, r, inv_d);
    const bool unc = far_pair_uncertain(tl, far_l, tr, far_r);
    if (unc) { far_l = slab_far_exact(lfx, lfy, lfz, r, inv_d, inv_lo); far_r = slab_far_exact(rfx, rfy, rfz, r, inv_d, inv_lo); }
    const bool hl_c = tl <= far_l && tl < maxd[i] && far_l > 0, hr_c = tr <= far_r && tr < maxd[i] && far_r > 0;
    float dl = RT_MISS_DIST, dr = RT_MISS_DIST;
    const bool hl = aabb_intersects(ld3(b), ld3(b + 3), r, maxd[i], dl);
    const bool hr = aabb_intersects(ld3(b + 6), ld3(b + 9), r, maxd[i], dr);
    o[1] = unc; o[2] = hl_c; o[3] = hr_c; o[4] = (hl_c ? tl : RT_MISS_DIST) > (hr_c ? tr : RT_MISS_DIST);
    o[5] = hl; o[6] = hr; o[7] = dl > dr;
}

extern "C" int rt_probe_boxpair_certified(int device, size_t n, const float* boxes, const float* rays, const float* max_dist, int32_t* out) {
    if (!boxes || !rays || !max_dist || !out) return rt_fail(RT_ERR_INVALID, "rt_probe_boxpair_certified: null argument");
    if (n == 0) return RT_OK;
    int rc = select_device(device);
    if (rc != RT_OK) return rc;
    DevBuf b, r, m, o;
    UP(b, boxes, n * 48); UP(r, rays, n * 24); UP(m, max_dist, n * 4);
    HIP_TRY(o.alloc(n * 32));
    probe_boxpair_certified_kernel<<<PROBE_GRID(n)>>>(n, b.as<float>(), r.as<float>(), m.as<float>(), o.as<int32_t>());
    FINISH();
    DOWN(out, o, n * 32);
    return RT_OK;
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

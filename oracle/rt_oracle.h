/*
 * rt_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the per-pixel sample loop of
 * SuperCat908809/Ray-Tracing-v06, each function citing the reference
 * file:line it follows (paths relative to /root/reference/).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker.  The product (librt06.so) never links,
 * loads or calls it.
 *
 * PIN STATUS.  The reference is CUDA C++.  What of it compiles in this image with plain g++ (NVIDIA's own <cuda_runtime.h> ships
 * inside the image's triton wheel; no stand-in headers are written) is compiled from where it lies into oracle/_ref/ and used
 * to generate committed fixtures, against which this file is checked bit for bit (tests/test_oracle_golden.py,
 * tests/test_reference_pins.py):
 *   - vendored GLM 0.9.9.7 + main/src/utilities/glm_utils.h            -> tests/golden/glm_*      every orc_glm_* function
 *   - rt_engine/geometry/aabb.cuh                                       -> ref_aabb_*, ref_aabbmisc_*  aabb_intersects, box_* helpers
 *   - rt_engine/geometry/HittableList.cuh, bvh_node.cuh (probe leaves)  -> ref_agg_*              list_/tree_closest_intersection
 *   - rt_engine/shaders/cu_Textures.cuh                                 -> ref_checker_*          checker_value
 *   - rt_engine/ray_data.cuh, geometry/BVH.cuh (layouts)                -> ref_ray_*, ref_layout.json
 * PARITY UNPINNED against an executing reference — the files do not build here (cuError.h needs <format>, cuda_utils.cuh
 * and Renderer.cu contain <<<>>>, cuRandom.cuh needs curand_kernel.h) and the reference ships no golden vectors (its one
 * gtest depends on cuRAND's host stream): _sphere_closest_intersection and the sphere hittables, BVH::ClosestIntersection
 * and the builders, Scatter, the cameras, sample_world, render_kernel, the scene generators — restated by hand from the
 * source text.  Their outputs of today are frozen in tests/golden/frozen_*.npz (oracle/gen_frozen.py) so that oracle and
 * kernels cannot drift together unnoticed.  The RNG is the build's own counter-seeded generator (the reference's cuRAND
 * XORWOW streams are not reproducible offline); Philox is pinned against the Random123 known-answer vectors.
 *
 * Arithmetic contract (shared with the HIP kernels): IEEE fp32, no FMA
 * contraction (-ffp-contract=off), correctly rounded / and sqrtf, GLM's
 * min/max/clamp NaN semantics, GLM's evaluation order; the single deviation
 * is powf(1-cos,5) (cu_materials.cuh:103) computed as x2=x*x; x4=x2*x2;
 * x5=x4*x because libm / OCML / CUDA powf differ in the last ulp.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z; } orc_v3;

/* Same byte layout as rt_bvh_node / rt_prim / rt_material / rt_camera in
 * include/rt06.h (restated here; the oracle includes nothing of the product). */
typedef struct { float min[3]; float max[3]; int32_t left; int32_t right; } orc_node;
typedef struct { float c0[3]; float radius; float c1[3]; uint32_t mat; } orc_prim;
typedef struct { float albedo[3]; float param; float albedo2[3]; uint32_t type; } orc_material;
/* quad(Q,u,v) of "Ray Tracing: The Next Week" (absent from the reference: SURVEY.md §8f rank 1); normal, D, w are
 * the cached plane quantities of that book's quad class */
typedef struct { float Q[3]; float D; float u[3]; uint32_t mat; float v[3]; float pad0; float normal[3]; float pad1; float w[3]; float pad2; } orc_quad;
typedef struct {
    uint32_t kind; int32_t root; uint32_t n_nodes, n_prims, n_materials, max_stack;
    float bounds_min[3], bounds_max[3];
    const orc_node* nodes; const orc_prim* prims; const orc_material* materials;
    /* extension beyond the reference (quads, emission, constant background): primitive index i >= n_prims is quad i - n_prims */
    const orc_quad* quads; uint32_t n_quads;
    uint32_t background;        /* 0: the reference's sky gradient (Renderer.cu:150-151); 1: constant background_color */
    float background_color[3];
    /* second extension (constant media need nothing here): Perlin tables and one RGB8 image for the textured materials */
    uint32_t image_width;
    const struct orc_perlin* perlin;
    const uint8_t* image;
    uint32_t image_height;
    uint32_t traversal;         /* BVH worlds: 0 = the live depth-first stack (BVH.cu:54-106), 1 = the reference's disabled distance-sorted queue (BVH.cu:17-49), off-by-one fixed, 2 = a 4-wide walk of the same tree (two levels per visit) */
} orc_world;
/* perlin::randvec / perm_x,y,z of "The Next Week" */
typedef struct orc_perlin { float randvec[256][3]; int32_t perm[3][256]; } orc_perlin;
typedef struct {
    uint32_t type; float o[3], u[3], v[3], w[3];
    float viewport_width, viewport_height, lens_radius, focus_dist, t0, t1;
} orc_camera;

typedef struct { float o[3]; float d[3]; float time; } orc_ray;

/* instrumented counts for the algorithmic-bytes model */
typedef struct {
    uint64_t samples, rays, box_tests, leaf_tests, shaded_hits, rng_draws;
    uint32_t max_stack;
} orc_counters;

/* ---- GLM vocabulary (pinned by tests/golden/glm_*.f32) ---- */
float  orc_glm_dot(const float a[3], const float b[3]);
void   orc_glm_cross(const float a[3], const float b[3], float out[3]);
void   orc_glm_normalize(const float a[3], float out[3]);
void   orc_glm_reflect(const float i[3], const float n[3], float out[3]);
void   orc_glm_refract(const float i[3], const float n[3], float eta, float out[3]);
void   orc_glm_mix3(const float a[3], const float b[3], float t, float out[3]);
float  orc_glm_mix1(float a, float b, float t);
void   orc_glm_min3(const float a[3], const float b[3], float out[3]);
void   orc_glm_max3(const float a[3], const float b[3], float out[3]);
float  orc_glm_compmax(const float a[3]);
float  orc_glm_compmin(const float a[3]);
void   orc_glm_clamp01_sqrt(const float a[3], float out[3]);
int    orc_glm_near_zero(const float a[3]);
float  orc_glm_length2(const float a[3]);
void   orc_glm_lerp(const float a[3], const float b[3], float t, float out[3]);
float  orc_glm_radians(float deg);

/* deterministic fp32 log / sin / acos / atan2 of the extension materials (not GLM, not libm: see rt_oracle.c) */
float orc_math_log(float x);
float orc_math_sin(float x);
float orc_math_acos(float x);
float orc_math_atan2(float y, float x);
void  orc_math_batch(int fn, size_t n, const float* a, const float* b, float* out);

/* ---- RNG ---- */
void   orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void   orc_rng_uniforms(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t stream,
                        uint32_t n, float* out);

/* ---- per-function batch entry points (twins of the rt_probe_* device probes) ---- */
void orc_aabb_batch(size_t n, const float* boxes, const float* rays, const float* max_dist,
                    int32_t* out_hit, float* out_dist);
void orc_sphere_batch(size_t n, const float* rays, const float* spheres, float* out_t);
int  orc_trace_batch(const orc_world* w, size_t n, const float* rays, int32_t* out_hit, float* out_t,
                     int32_t* out_prim, float* out_normal);
void orc_scatter_batch(uint64_t seed, size_t n, const orc_material* mats, const float* rays,
                       const float* dist, const float* normals, const uint32_t* keys,
                       int32_t* out_scattered, float* out_rays, float* out_atten, uint32_t* out_draws);
void orc_camera_batch(uint64_t seed, const orc_camera* cam, size_t n, const float* st,
                      const uint32_t* keys, float* out_rays, uint32_t* out_draws);
int  orc_radiance_batch(const orc_world* w, const orc_camera* cam, uint32_t width, uint32_t height,
                        uint32_t max_depth, uint64_t seed, size_t n, const uint32_t* keys,
                        float* out_radiance);
void orc_sphere_index(const orc_camera* cam, uint32_t width, uint32_t height, size_t n_spheres,
                      const float* spheres, int32_t* out_index);

/* twins of the fixtures generated from the reference's own aabb / texture / ray_data headers (tests/golden/ref_*) */
void orc_aabb_misc_batch(size_t n, const float* boxes, float* out);
void orc_checker_batch(size_t n, const float* in, float* out);
void orc_ray_batch(size_t n, const float* in, float* out);
int  orc_trace_counts(const orc_world* w, size_t n, const float* rays, uint32_t* out_leaf_tests, uint32_t* out_box_tests);

/* ---- cameras (constructors) ---- */
void orc_camera_pinhole(const float lookfrom[3], const float lookat[3], const float up[3],
                        float vfov, float aspect, orc_camera* out);
void orc_camera_defocus(const float lookfrom[3], const float lookat[3], const float up[3],
                        float vfov, float aspect, float aperture, float focus_dist, orc_camera* out);
void orc_camera_motion(const float lookfrom[3], const float lookat[3], const float up[3],
                       float vfov, float aspect, float t0, float t1, orc_camera* out);

/* ---- the render loop (render_kernel + sample_world) ---- */
/* Renders rows [0,height) with n_threads pthreads (dynamic row queue).
 * out_rgba: width*height*4 floats, row 0 = bottom.  counters may be NULL.
 * Returns 0, or 4 if the traversal stack (32, BVH.cu:17) would overflow.   */
int orc_render(const orc_world* w, const orc_camera* cam, uint32_t width, uint32_t height,
               uint32_t spp, uint32_t max_depth, uint64_t seed, int n_threads, float* out_rgba,
               orc_counters* counters);

int orc_render_pixels(const orc_world* w, const orc_camera* cam, uint32_t width, uint32_t height, uint32_t spp,
                      uint32_t max_depth, uint64_t seed, size_t n, const uint32_t* gids, float* out_rgba);

/* ---- scene generation + BVH build (host side of the path, a15/a16) ---- */
typedef struct orc_scene orc_scene;
orc_scene* orc_scene_book1_final(uint64_t seed);
orc_scene* orc_scene_book2_moving(uint64_t seed);
orc_scene* orc_scene_three_spheres(void);
/* generic: build from caller arrays; builder 0 = top-down median, 1 = binned SAH,
 * 2 = bottom-up, 3 = HittableList                                              */
orc_scene* orc_scene_from_arrays(size_t n_prims, const orc_prim* prims, size_t n_mats,
                                 const orc_material* mats, int builder);
/* same with quads (Q,u,v,mat given; cached plane quantities are recomputed) and a background */
orc_scene* orc_scene_from_arrays_ext(size_t n_prims, const orc_prim* prims, size_t n_quads, const orc_quad* quads,
                                     size_t n_mats, const orc_material* mats, int builder, uint32_t background,
                                     const float background_color[3]);
/* the Cornell box of "The Next Week" (BASELINE.json configs[3]): 5 walls, a light, two rotated boxes as 12 quads */
orc_scene* orc_scene_cornell_box(void);
orc_scene* orc_scene_book2_final(uint64_t seed);
void orc_scene_world(const orc_scene* s, orc_world* out);
/* perlin::perlin() from the build's host stream (id 0x9E81) / the image of image_texture, attached to a scene */
void orc_scene_set_perlin(orc_scene* s, uint64_t seed);
void orc_scene_set_image(orc_scene* s, uint32_t width, uint32_t height, const uint8_t* rgb);
void orc_scene_free(orc_scene* s);

#ifdef __cplusplus
}
#endif
#endif

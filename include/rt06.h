/*
 * rt06.h — C ABI of the MI355X-native path-tracer hot path.
 *
 * This is the drop-in boundary for the per-pixel sample loop of
 * SuperCat908809/Ray-Tracing-v06.  The reference has no FFI layer; the boundary
 * the hot path sits behind is its C++ class `Renderer`
 * (main/src/Renderer.h:38-46) plus the scene vocabulary that produces the
 * `world` argument.  Each entry point below names the reference interface it
 * replaces.  The reference-shaped C++ classes in include/rt06/ (Renderer,
 * SphereHandle, BVH_Handle::Factory, cameras, materials, scenes) are thin
 * header-only wrappers over exactly these functions.
 *
 * Conventions
 *  - plain pointers and sizes only; every function returns an int status
 *    (0 = RT_OK) and never throws; rt_last_error() returns the message of the
 *    last failure on the calling thread.  The reference's CUDA_ASSERT is a
 *    no-op in Release (utilities/cuda_utilities/cuError.h:25-29); this ABI is
 *    never silent.
 *  - all floating point is IEEE fp32; vectors are float[3] (x,y,z).
 *  - the framebuffer is row-major float RGBA, 16 B per pixel, alpha = 1,
 *    row 0 = BOTTOM row of the image, values sqrt-gamma in [0,1]
 *    (Renderer.cu:206-216).
 *  - objects are not thread-safe; distinct objects are independent.
 */
#ifndef RT06_H
#define RT06_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_ERR_INVALID 1   /* bad argument / malformed scene                */
#define RT_ERR_HIP 2       /* a HIP runtime call failed                     */
#define RT_ERR_NO_DEVICE 3 /* no usable gfx950 device                       */
#define RT_ERR_STACK 4     /* BVH deeper than the traversal stack allows    */

/* ------------------------------------------------------------------ */
/* GPU-linear scene records (what hittable_list / bvh_node / BVH flatten to) */
/* ------------------------------------------------------------------ */

/* == BVH::Node, rt_engine/geometry/BVH.cuh:16-25 (32 B, same field order).
 * RT_WORLD_BVH      : left == -1  -> leaf, right = primitive index;
 *                     otherwise left/right are child node indices.
 * RT_WORLD_NODE_TREE: (== bvh_node, rt_engine/geometry/bvh_node.cuh:9-25)
 *                     left/right are child references: r >= 0 is a node
 *                     index, r < 0 is primitive index (-r - 1).            */
typedef struct rt_bvh_node {
    float   min[3];
    float   max[3];
    int32_t left;
    int32_t right;
} rt_bvh_node;

#define RT_PRIM_MOVING 0x80000000u
/* Sphere (SphereHittable.cuh:35-52) / MovingSphere (:70-87) + its material
 * binding (SphereHittable{sphere*,mat*}, :56-67), 32 B.
 * mat = material index, bit 31 set for a MovingSphere (c1 is then used).   */
typedef struct rt_prim {
    float    c0[3];
    float    radius;
    float    c1[3];
    uint32_t mat;
} rt_prim;

/* ---- beyond the reference (the reference has no quads, no emission, no background colour:
 * only the commented `accum_radiance` placeholders of Renderer.cu:142,152,157,163,179).  Semantics follow
 * "Ray Tracing: The Next Week" (quad(Q,u,v), diffuse_light, camera background) in the reference's conventions.
 * PARITY UNPINNED: checked GPU against the build's own CPU oracle only. ---- */
/* quad(Q,u,v,mat): plane point Q, edge vectors u and v; normal/D/w are the cached plane quantities
 * (n = cross(u,v), normal = unit(n), D = dot(normal,Q), w = n/dot(n,n)), filled by rt_scene_add_quad.  80 B.     */
typedef struct rt_quad {
    float    Q[3];      float D;
    float    u[3];      uint32_t mat;
    float    v[3];      float pad0;
    float    normal[3]; float pad1;
    float    w[3];      float pad2;
} rt_quad;

enum {
    RT_MAT_LAMBERTIAN = 0,         /* LambertianAbstract  cu_materials.cuh:44-65  param unused          */
    RT_MAT_METAL = 1,              /* MetalAbstract       cu_materials.cuh:68-96  param = fuzz          */
    RT_MAT_DIELECTRIC = 2,         /* DielectricAbstract  cu_materials.cuh:106-144 param = ior          */
    RT_MAT_LAMBERTIAN_CHECKER = 3, /* LambertianTexture   cu_materials.cuh:16-41  albedo/albedo2 = even/odd colour, param = 1/scale */
    RT_MAT_DIFFUSE_LIGHT = 4,      /* diffuse_light of "The Next Week" (not in the reference): emits albedo, never scatters    */
    RT_MAT_ISOTROPIC = 5,          /* isotropic phase function of "The Next Week" (not in the reference): a SPHERE with this
                                    * material is a constant_medium bounded by it; albedo = colour, param = density          */
    RT_MAT_LAMBERTIAN_NOISE = 6,   /* lambertian(noise_texture(scale)) of "The Next Week" (not in the reference): colour =
                                    * albedo * (1 + sin(param * p.z + 10 * turb(p, 7))) over the world's Perlin tables        */
    RT_MAT_LAMBERTIAN_IMAGE = 7    /* lambertian(image_texture) of "The Next Week": colour = the world's RGB8 image at (u, v) — on a sphere
                                    * sphere::get_sphere_uv of the normal, on a quad the planar coordinates (alpha, beta) of quad::hit   */
};
typedef struct rt_material {
    float    albedo[3];
    float    param;
    float    albedo2[3];
    uint32_t type;
} rt_material;

enum {
    RT_WORLD_BVH = 0,       /* flat index-linked BVH          (BVH.cu:54-106)           */
    RT_WORLD_LIST = 1,      /* HittableList                   (HittableList.cuh:21-34)  */
    RT_WORLD_NODE_TREE = 2  /* pointer-recursive bvh_node     (bvh_node.cuh:19-24)      */
};
/* The `const Hittable* d_world_ptr` argument of Renderer::MakeRenderer,
 * resolved to flat host arrays.  Borrowed during rt_renderer_create only.   */
/* Perlin noise tables of "The Next Week" (perlin::randvec, perm_x/y/z), generated by rt_scene_set_perlin; 6144 B */
typedef struct rt_perlin {
    float   randvec[256][3];
    int32_t perm[3][256];
} rt_perlin;

typedef struct rt_world_flat {
    uint32_t kind;         /* RT_WORLD_*                                              */
    int32_t  root;         /* root node index (BVH: last node; NODE_TREE: child ref)  */
    uint32_t n_nodes;
    uint32_t n_prims;
    uint32_t n_materials;
    uint32_t max_stack;    /* traversal-stack bound derived from the tree at build time */
    float    bounds_min[3];/* world bounds (HittableList pre-test, HittableList.cuh:22) */
    float    bounds_max[3];
    const rt_bvh_node* nodes;
    const rt_prim*     prims;
    const rt_material* materials;
    /* extension (see rt_quad): a primitive index i >= n_prims means quad i - n_prims                      */
    const rt_quad*     quads;
    uint32_t n_quads;
    uint32_t background;         /* 0: the reference's sky gradient (Renderer.cu:150-151); 1: background_color */
    float    background_color[3];
    uint32_t image_width;        /* RT_MAT_LAMBERTIAN_IMAGE: one RGB8 image per world, row 0 = top (as stb_image loads it) */
    const rt_perlin* perlin;     /* RT_MAT_LAMBERTIAN_NOISE: the world's noise tables, or NULL                            */
    const uint8_t*   image;      /* image_width * image_height * 3 bytes, or NULL                                         */
    uint32_t image_height;
    uint32_t traversal;          /* RT_WORLD_BVH only: RT_TRAVERSAL_STACK (0, the live path), RT_TRAVERSAL_QUEUE (1) or RT_TRAVERSAL_WIDE4 (2) */
} rt_world_flat;                 /* 128 B */

/* How BVH::ClosestIntersection walks the tree (rt_scene_set_traversal).
 * STACK: the reference's live depth-first walk, near child first (BVH.cu:54-106, `_USE_PRIO_QUEUE false`) — every streaming kernel.
 * QUEUE: the distance-sorted queue the reference carries but disables (BVH.cu:17-49, :80-86): best-first over the whole frontier,
 *        with its off-by-one (`distances[head]` written after `head++`, :37-39) FIXED.  Capacity 32 (_PRIO_QUEUE_ELEM_COUNT) is
 *        checked: an overflow is RT_ERR_STACK at the next synchronising call, never silent.  Renders on the streaming kernel
 *        (variant 0 / 2: its queue mode — a lane walks its whole trace with the queue when the trace begins; the framebuffer is
 *        the oracle's bit for bit) and on the baseline kernel (variant 1); the stack-walking variants 3-5 refuse it.  On the
 *        Book-1 final scene it saves 0.6 % of the box tests and costs 4.8 % more leaf tests (instrumented oracle); its frontier
 *        is one sorted list per ray, so it cannot share the wave-level hot loop: 1.29 against 5.6 Gsamples/s (EXPERIMENTS.md E2).
 * WIDE4: a 4-wide walk of the SAME binary tree (not in the reference, whose nodes are binary, BVH.cuh:16-25): a visit looks
 *        two levels down — up to four grandchild boxes, tested against rec.distance, nearest first, pushed far-to-near, culling at push time
 *        only (BVH.cu:87-96's rule generalised); the intermediate children's boxes are not tested.  Its 32 entries (at most 3 * ceil(depth / 2) + 1
 *        are needed) are checked like the queue's: RT_ERR_STACK at the next synchronising call.  Oracle twin: orc_world.traversal == 2.  Renders where the queue renders (streaming kernel's
 *        lane-walk mode bit-identical to the oracle, baseline kernel); measured next to the binary walk in EXPERIMENTS.md E3.               */
enum { RT_TRAVERSAL_STACK = 0, RT_TRAVERSAL_QUEUE = 1, RT_TRAVERSAL_WIDE4 = 2 };

enum {
    RT_CAM_PINHOLE = 0,  /* PinholeCamera     cu_Cameras.cuh:12-31 */
    RT_CAM_DEFOCUS = 1,  /* DefocusBlurCamera cu_Cameras.cuh:34-65 */
    RT_CAM_MOTION = 2    /* MotionBlurCamera  cu_Cameras.cuh:68-90 */
};
/* Camera POD, passed by value to the kernel like LaunchParams::cam
 * (Renderer.cu:99-108,117).  For PINHOLE/MOTION u,v are pre-scaled by the
 * viewport; for DEFOCUS they are unit vectors and viewport_* are separate.  */
typedef struct rt_camera {
    uint32_t type;
    float o[3], u[3], v[3], w[3];
    float viewport_width, viewport_height;
    float lens_radius, focus_dist;
    float t0, t1;
} rt_camera;

const char* rt_last_error(void);

/* ------------------------------------------------------------------ */
/* cameras — constructors of cu_Cameras.cuh:16-28, 40-52, 73-85        */
/* ------------------------------------------------------------------ */
int rt_camera_pinhole(const float lookfrom[3], const float lookat[3], const float up[3],
                      float vfov, float aspect, rt_camera* out);
int rt_camera_defocus(const float lookfrom[3], const float lookat[3], const float up[3],
                      float vfov, float aspect, float aperture, float focus_dist, rt_camera* out);
int rt_camera_motion(const float lookfrom[3], const float lookat[3], const float up[3],
                     float vfov, float aspect, float time0, float time1, rt_camera* out);

/* ------------------------------------------------------------------ */
/* scene construction — host side, replaces SphereHandle / newOnDevice /
 * BVH_Handle::Factory / HittableList / bvh_node / SceneBook2BVH::Factory  */
/* ------------------------------------------------------------------ */
typedef struct rt_scene rt_scene;

int rt_scene_create(rt_scene** out);
void rt_scene_destroy(rt_scene* s);

/* newOnDevice<LambertianAbstract<G>>(albedo) etc. (Scenes.cu:222,235,242,248).
 * Returns the material index in *out_id.                                    */
int rt_scene_add_material(rt_scene* s, uint32_t type, const float albedo[3], float param,
                          const float albedo2[3], int32_t* out_id);
/* SphereHandle::MakeSphere / MakeMovingSphere (SphereHittable.cuh:134-154).
 * Returns the primitive index; bounds as getSphereBounds/getMovingSphereBounds
 * (SphereHittable.cu:52-54, 85-89).                                          */
int rt_scene_add_sphere(rt_scene* s, const float center[3], float radius, int32_t mat, int32_t* out_prim);
int rt_scene_add_moving_sphere(rt_scene* s, const float c0[3], const float c1[3], float radius,
                               int32_t mat, int32_t* out_prim);
int rt_scene_prim_bounds(const rt_scene* s, int32_t prim, float out_min[3], float out_max[3]);
/* quad(Q,u,v,mat) of "The Next Week"; returns the quad index (worlds: BVH builders and HittableList, where the
 * quads follow the spheres; bvh_node trees take spheres only).                                            */
int rt_scene_add_quad(rt_scene* s, const float Q[3], const float u[3], const float v[3], int32_t mat, int32_t* out_quad);
/* camera::background of "The Next Week": mode 0 = the reference's sky gradient, 1 = constant colour       */
int rt_scene_set_background(rt_scene* s, uint32_t mode, const float color[3]);
/* selects RT_TRAVERSAL_STACK / RT_TRAVERSAL_QUEUE / RT_TRAVERSAL_WIDE4 for the BVH world of this scene (see the enum) */
int rt_scene_set_traversal(rt_scene* s, uint32_t mode);
/* perlin::perlin() of "The Next Week": 256 random unit vectors + three Fisher-Yates permutations, drawn from the
 * build's host stream (rt_host_uniforms, stream id 0x9E81) with this seed                                  */
int rt_scene_set_perlin(rt_scene* s, uint64_t seed);
/* the image of image_texture (the book loads earthmap.jpg; any RGB8 array here), copied                     */
int rt_scene_set_image(rt_scene* s, uint32_t width, uint32_t height, const uint8_t* rgb);

/* BVH_Handle::Factory::BuildBVH_TopDown -> _build_bvh_rec1 (BVH.cu:166-210):
 * median split on the longest axis, leaf size 1, post-order numbering, root =
 * last node.  Reorders the primitives (hittables[] = sorted order, :174-177). */
int rt_scene_build_bvh_topdown(rt_scene* s);
/* _build_bvh_rec2 + _find_optimal_split + _partition_by_split (BVH.cu:212-304) */
int rt_scene_build_bvh_sah(rt_scene* s);
/* BuildBVH_BottomUp (BVH.cu:315-384), O(n^3) agglomerative                    */
int rt_scene_build_bvh_bottomup(rt_scene* s);
/* HittableList(objects, count, bounds) (HittableList.cuh:19)                  */
int rt_scene_set_world_list(rt_scene* s);
/* bvh_node(left,right,bounds) (bvh_node.cuh:17).  Child refs: >= 0 node
 * returned earlier, < 0 primitive (-prim - 1).  bounds may be NULL = union of
 * the children's bounds.                                                      */
int rt_scene_add_bvh_node(rt_scene* s, int32_t left_ref, int32_t right_ref,
                          const float bmin[3], const float bmax[3], int32_t* out_ref);
int rt_scene_set_world_node_tree(rt_scene* s, int32_t root_ref);

/* SceneBook2BVH::getWorldPtr (Scenes.h:72) resolved to flat arrays.  Pointers
 * stay valid until the scene is modified or destroyed.                        */
int rt_scene_get_flat(const rt_scene* s, rt_world_flat* out);

/* cuHostRND::next (utilities/cuda_utilities/cuHostRND.h:9-32, cuHostRND.cpp:57-65): the host uniform
 * stream scene factories draw from.  Uniforms first .. first+n-1 of the library's counter-based host
 * stream for `seed` (each in (0,1]); stateless, so no generator object is needed.                      */
int rt_host_uniforms(uint64_t seed, uint32_t first, uint32_t n, float* out);

/* Prefab scenes.  The reference draws its layout from cuRAND's host XORWOW
 * stream (cuHostRND, seed 1984), which cannot be reproduced without cuRAND;
 * these use the library's own counter-based host stream with the reference's
 * draw pattern (Scenes.cu:229-252; test.cpp:36-62).
 *  book1_final   : 488 static spheres + BVH   (disabled SceneBook1, Scenes.cu:57-115)
 *  book2_moving  : moving Lambertians + BVH   (live SceneBook2BVH, Scenes.cu:219-270)
 *  three_spheres : Book-1 three-spheres scene as a HittableList (config 1)    */
int rt_scene_book1_final(uint64_t seed, rt_scene** out);
int rt_scene_book2_moving(uint64_t seed, rt_scene** out);
int rt_scene_three_spheres(rt_scene** out);
/* BASELINE.json configs[3]: the Cornell box of "The Next Week" (5 walls, light, two rotated boxes = 18 quads),
 * black background, median-split BVH.  Not in the reference (no quads / emission there).                  */
int rt_scene_cornell_box(rt_scene** out);
/* box(a, b, mat) of "The Next Week" as 6 quads, rotated about y and translated on the host (the book wraps instances) */
int rt_scene_add_box(rt_scene* s, const float a[3], const float b[3], int32_t mat, float rotate_y_degrees,
                     const float translate[3], int32_t* out_first_quad);
/* BASELINE.json configs[4]: final_scene() of "The Next Week" — 2401 quads, 1008 spheres, two constant media, a marble
 * and an image texture (a synthetic planet stands in for earthmap.jpg), black background, median-split BVH.  Not in the
 * reference.  Camera of the book: lookfrom (478,278,-600), lookat (278,278,0), vfov 40, time 0..1.               */
int rt_scene_book2_final(uint64_t seed, rt_scene** out);

/* ------------------------------------------------------------------ */
/* Renderer — main/src/Renderer.h:38-46                                */
/* ------------------------------------------------------------------ */
typedef struct rt_renderer rt_renderer;

typedef struct rt_render_config {
    uint32_t width, height;       /* Renderer::MakeRenderer args 1-2 */
    uint32_t samples_per_pixel;   /* arg 3 */
    uint32_t max_depth;           /* arg 4 */
    uint64_t seed;                /* reference hard-codes 1984 (Renderer.cu:51) */
    int32_t  device;              /* HIP device ordinal */
    /* tile sharding across the GPUs of one node: this renderer owns the 8x8
     * pixel tiles t with t % world_size == rank (row-major tile order).      */
    uint32_t rank, world_size;
    uint32_t variant;             /* 0 = default (3 where the world allows it, else 2, else 1); 1 baseline wave-per-pixel kernel,
                                     2 streaming kernel with verbatim box tests (IEEE divisions), 3 = 2 + exact division without
                                     dividing (BVH worlds with box coordinates in [2^-40, 2^40)), 4 = 3 + filtered predicates
                                     (experimental, reference features only), 5 = 3 with rays exchanged between tracer and
                                     shader waves of a workgroup through LDS rings (render_kernel_xchg; LDS-resident BVH worlds
                                     of the reference's feature set; measured slower than 3, kept as an opt-in: EXPERIMENTS.md),
                                     6 = 3 in TOLERANCE MODE (opt-in, never chosen by 0): the box tests' plane parameters are
                                     (b - o) * RN(1/d) instead of aabb.cuh:30-31's quotients — inside BASELINE.json's |delta| < 1e-3,
                                     NOT bit-exact by construction (measured: 0 differing pixels on BASELINE configs[1..2] at full
                                     size, dominant kernel 1.24-1.26x faster); LDS-resident RT_WORLD_BVH worlds of the reference's own
                                     feature set only — worlds with quads / lights / media are refused (a quad's edges are its box's
                                     edges: the Cornell box at 5000 spp left the tolerance in one pixel, EXPERIMENTS.md E4).
                                     Same image bits for 2..5; worlds beyond the LDS take the global-memory form of 2 / 3
                                     (rt_renderer_kernel_info).                                                                */
} rt_render_config;

/* Renderer::MakeRenderer (Renderer.cu:31-67).  Copies the flat world and the
 * camera; allocates the device framebuffer.  No RNG-state array is needed
 * (counter-based RNG), so init_random_states (Renderer.cu:22-29) has no twin. */
int rt_renderer_create(const rt_render_config* cfg, const rt_camera* cam,
                       const rt_world_flat* world, rt_renderer** out);
void rt_renderer_destroy(rt_renderer* r);

/* Renderer::Render (Renderer.cu:111-137): blocking; launches on the
 * renderer's own stream and waits.                                          */
int rt_renderer_render(rt_renderer* r);
/* Same launch on a caller-provided hipStream_t, no host synchronisation.
 * d_out = device buffer for this rank's shard (rt_renderer_shard_floats
 * floats) or NULL to use the renderer's own framebuffer.                    */
int rt_renderer_render_async(rt_renderer* r, void* hip_stream, float* d_out);
/* HIP-event time of the last render launch(es) in ms (cudaTimer twin,
 * Renderer.cu:127-136).  Synchronises on the events.                        */
int rt_renderer_last_kernel_ms(rt_renderer* r, float* out_ms);
/* Per-kernel HIP-event times (ms) of one of the last 32 render calls (renders_back = 0: the most recent), measured on the
 * stream the kernels ran on: out[0] = primary_rays_kernel, out[1] = the dominant kernel (render_kernel_stream /
 * render_kernel_xchg), out[2] = resolve_kernel — each SUMMED over the passes of that call (rt_renderer_pass_info).
 * Synchronises on that call's end.  Streaming variants (>= 2) only.                                                       */
int rt_renderer_kernel_times(rt_renderer* r, uint32_t renders_back, float out_ms[3]);
/* How a render is cut into passes: out[0] = passes per render, out[1] = samples per pixel per pass, out[2] = HBM bytes per
 * sample index of a pass (12 B radiance + 48 B primary-ray record), out[3] = bytes of the per-pass buffers this renderer
 * holds (sample buffer + primary rays + running sums).  A pass is sized by a budget over ALL of those buffers: 120 GiB by
 * default but at most 45 % of the HBM that is free when the renderer is created (RT06_PASS_BUDGET_BYTES to change it, RT06_PASS_SPP to
 * force the samples per pixel per pass: tests); if the device cannot provide the buffers the passes are halved until it can.           */
int rt_renderer_pass_info(rt_renderer* r, uint64_t out[4]);
/* Which kernel the renderer resolved to: out[0] = variant actually used (1..6), out[1] = 1 when the scene image is
 * LDS-resident (0: baseline kernel, or a world too large for the LDS, served from global memory / L2 with 32-bit
 * references), out[2] = workgroup size, out[3] = workgroups per CU.                                              */
int rt_renderer_kernel_info(rt_renderer* r, uint32_t out[4]);
/* Renderer::DownloadRenderbuffer (Renderer.cu:94-96): width*height*4 floats,
 * row-major, row 0 = bottom.  Only valid for world_size == 1.               */
int rt_renderer_download(rt_renderer* r, float* host_rgba, size_t n_floats);
/* Number of floats in this rank's compact shard (n_local_tiles * 64 * 4);
 * identical on every rank.                                                  */
int rt_renderer_shard_floats(const rt_renderer* r, size_t* out);
/* Rank-0 side of the frame-end gather: `d_gathered` holds world_size shards
 * back to back (rank-major); writes the row-major width*height*4 image.     */
int rt_renderer_assemble(rt_renderer* r, const float* d_gathered, float* d_image, void* hip_stream);

/* ------------------------------------------------------------------ */
/* Multi-GPU renderer — the same three entry points (Renderer.h:38-46)  */
/* over the N GPUs of one node, driven by ONE host process.            */
/* ------------------------------------------------------------------ */
/* The reference is single-GPU.  Rank i = devices[i] (NULL: 0 .. n_gpus-1) renders the 8x8 tiles t with
 * t % n_gpus == i; ONE grouped RCCL exchange over xGMI (ncclSend from every rank, ncclRecv on rank 0) gathers the shards
 * on devices[0] at frame end and a de-interleave kernel assembles the row-major frame there.  cfg->device / rank /
 * world_size are ignored.  The image has the same bits for every n_gpus.  RCCL (librccl.so.1) is bound at first use.
 * RT06_MULTI_TRANSPORT=memcpy (tests, single-GPU boxes) replaces the RCCL exchange by hipMemcpyAsync on the ranks' own
 * streams and lifts the one-rank-per-device rule (devices may repeat; NULL = i % device count), so that the N > 1 branch
 * — shard offsets, stream ordering, assembly, download — runs on ONE GPU, where RCCL refuses two ranks.                 */
typedef struct rt_multi_renderer rt_multi_renderer;
int rt_multi_renderer_create(const rt_render_config* cfg, const rt_camera* cam, const rt_world_flat* world,
                             uint32_t n_gpus, const int32_t* devices, rt_multi_renderer** out);
void rt_multi_renderer_destroy(rt_multi_renderer* m);
/* Renderer::Render: blocking; renders all shards side by side, gathers, assembles.                                      */
int rt_multi_renderer_render(rt_multi_renderer* m);
/* Renderer::DownloadRenderbuffer: width*height*4 floats from devices[0].                                                */
int rt_multi_renderer_download(rt_multi_renderer* m, float* host_rgba, size_t n_floats);
/* ms of the last render: out[0] host wall-clock of Render(), out[1] slowest rank's kernels (HIP events),
 * out[2] exchange + assembly on devices[0] (HIP events)                                                                 */
int rt_multi_renderer_times(rt_multi_renderer* m, float out_ms[3]);
int rt_multi_renderer_gpus(const rt_multi_renderer* m, uint32_t* out);
/* HOST (no GPU): the shard layout every rank uses — out = {tiles_x, n_tiles, n_local_tiles, shard_floats} — and the global
 * pixel id of every shard position of one rank (n = n_local_tiles * 64 entries, 0xffffffff for padding).                */
int rt_shard_layout(uint32_t width, uint32_t height, uint32_t world_size, uint32_t out[4]);
int rt_shard_pixel_map(uint32_t width, uint32_t height, uint32_t world_size, uint32_t rank, uint32_t* out_gid, size_t n);

/* ------------------------------------------------------------------ */
/* Device probes: run ONE hot-path function over an array of inputs on  */
/* the GPU.  Used by the parity tests (per-function golden vectors) —   */
/* the twin of google_testing/test.cpp's host-vs-device differential.   */
/* All pointers are HOST pointers; the probes copy in/out themselves.   */
/* ------------------------------------------------------------------ */
/* aabb::intersects (aabb.cuh:30-44): boxes n*6 (min,max), rays n*6 (o,d),
 * max_dist n -> hit n (0/1), dist n (only written when hit, else left 0).   */
int rt_probe_aabb(int device, size_t n, const float* boxes, const float* rays, const float* max_dist,
                  int32_t* out_hit, float* out_dist);
/* _sphere_closest_intersection (SphereHittable.cuh:15-33): rays n*6,
 * spheres n*4 (c,r) -> t n                                                   */
int rt_probe_sphere(int device, size_t n, const float* rays, const float* spheres, float* out_t);
/* Hittable::ClosestIntersection on a whole world: rays n*7 (o,d,time) ->
 * hit n, t n, prim n, normal n*3                                             */
int rt_probe_trace(int device, const rt_world_flat* world, size_t n, const float* rays,
                   int32_t* out_hit, float* out_t, int32_t* out_prim, float* out_normal);
/* Material::Scatter (cu_materials.cuh:52,77,115,27): per case a material,
 * in-ray n*7, hit distance n, outward normal n*3, RNG key (pixel,sample) n*2
 * -> scattered n (0/1), out ray n*7, attenuation n*3, RNG blocks consumed n  */
int rt_probe_scatter(int device, uint64_t seed, size_t n, const rt_material* mats, const float* rays,
                     const float* dist, const float* normals, const uint32_t* keys,
                     int32_t* out_scattered, float* out_rays, float* out_atten, uint32_t* out_draws);
/* camera sample_ray (cu_Cameras.cuh:27,54,87): st n*2, keys n*2 -> ray n*7, RNG blocks consumed n */
int rt_probe_camera(int device, uint64_t seed, const rt_camera* cam, size_t n, const float* st,
                    const uint32_t* keys, float* out_rays, uint32_t* out_draws);
/* one full sample (render_kernel body for one s + sample_world,
 * Renderer.cu:139-181,198-204): keys n*2 (pixel gid, sample) -> radiance n*3 */
int rt_probe_radiance(const rt_render_config* cfg, const rt_camera* cam, const rt_world_flat* world,
                      size_t n, const uint32_t* keys, float* out_radiance);
/* google_testing/test.cpp:112-135 `_sphere_index_ker`: brute-force nearest
 * sphere index per pixel, pinhole camera, NDC = x/(w-1)*2-1 (test.cpp:118-119) */
int rt_probe_sphere_index(int device, const rt_camera* cam, uint32_t width, uint32_t height,
                          size_t n_spheres, const float* spheres, int32_t* out_index);
/* raw uniforms of the counter-based RNG: keys n*2, n_draws each -> n*n_draws */
int rt_probe_rng(int device, uint64_t seed, size_t n, const uint32_t* keys, uint32_t n_draws, float* out);
/* the deterministic fp32 log / sin / acos / atan2 of the extension materials (fn = 0..3; b is the second atan2
 * argument, ignored otherwise): out[i] = f(a[i], b[i])                                                        */
int rt_probe_math(int device, int fn, size_t n, const float* a, const float* b, float* out);

/* The device math vocabulary (csrc/rt_math.hpp: GLM's dot / cross / normalize / reflect / refract / mix / min / max /
 * compMax / compMin / clamp+sqrt / radians and glm_utils.h's near_zero / length2 / linear_interpolate; fn = 0..15 in the
 * order of tests/golden/glm_*: dot cross normalize reflect refract mix3 mix1 min3 max3 compmax compmin clamp01_sqrt
 * near_zero length2 lerp radians) and fn 16 = Ray::at + isBackfacing (ray_data.cuh:14,44-46).  in = n * nin floats,
 * out = n * nout floats in the layout of those fixtures.  Run by the GPU tests on the reference-generated vectors.   */
int rt_probe_glm(int device, int fn, size_t n, const float* in, float* out);
/* HOST probe (no GPU): the aabb helpers of the BVH builders — longest_axis, surface_area, centeroid, union, +=,
 * box_{x,y,z}_compare (aabb.cuh:19,24,46-68,78-88): boxes n*12 (a.min a.max b.min b.max) -> out n*20.              */
int rt_probe_aabb_misc(size_t n, const float* boxes, float* out);

/* Verification probes for the fast exact division of the streaming kernel (csrc/rt_fastdiv.hpp).
 * rt_probe_aabb_regular: boxes n*6, rays n*6, max_dist n -> the "regular ray" classification n, and
 * hit/dist of the 5-instruction-division box test (only meaningful where regular == 1).             */
int rt_probe_aabb_regular(int device, size_t n, const float* boxes, const float* rays, const float* max_dist,
                          int32_t* out_regular, int32_t* out_hit, float* out_dist);
/* The filtered box-pair predicates of one inner-node visit (rt_fastdiv.hpp): boxes n*12 (left min,max,
 * right min,max), rays n*6, max_dist n -> out n*8 int32: [regular, uncertain, hit_left, hit_right, swap,
 * exact hit_left, exact hit_right, exact left_dist > right_dist].                                       */
int rt_probe_boxpair_filtered(int device, size_t n, const float* boxes, const float* rays, const float* max_dist, int32_t* out);
/* The box pair of the default hot loop (rt_fastdiv.hpp, CERTIFIED FAR PLANES: exact near parameters, far parameters as products with the
 * rounded reciprocal whose `tmin <= tmax` decisions are certified, exact redo otherwise) next to aabb::intersects (aabb.cuh:26-44) on both
 * boxes: same layout as above -> out n*8 int32: [regular, could not certify, hit_left, hit_right, left_dist > right_dist,
 * exact hit_left, exact hit_right, exact left_dist > right_dist].                                                                       */
int rt_probe_boxpair_certified(int device, size_t n, const float* boxes, const float* rays, const float* max_dist, int32_t* out);
/* Exhaustive self-test: for each of n_den divisor significands starting at first_den (0 .. 2^23-1) and
 * ALL 2^23 numerator significands, compare the 5-instruction quotient with IEEE n/d bit for bit.
 * num_exp / den_exp are the unbiased exponents given to numerator and divisor.  Returns the number of
 * mismatching pairs in *mismatches and one example in example[2] (numerator, divisor bits).          */
int rt_selftest_fastdiv(int device, uint32_t first_den, uint32_t n_den, int32_t num_exp, int32_t den_exp,
                        uint64_t* mismatches, uint32_t example[2]);

/* The same sweep for the 4-instruction quotient from a two-word reciprocal (fast_div_exact4, the one the default
 * kernel uses).                                                                                             */
int rt_selftest_fastdiv4(int device, uint32_t first_den, uint32_t n_den, int32_t num_exp, int32_t den_exp,
                         uint64_t* mismatches, uint32_t example[2]);
/* Exhaustive self-test of the 3-instruction reciprocal used for RN(1/d) of rays in the fast-division class: every fp32
 * x with 2^-40 <= |x| < 2^40 (1,342,177,280 values) against the IEEE division 1.0f / x, bit for bit.            */
int rt_selftest_fastrcp(int device, uint64_t* checked, uint64_t* mismatches, uint32_t* example);

/* library / device info */
int rt_device_count(int* out);
/* out = {compute units, peak engine clock in kHz, device memory in MiB, memory clock in kHz} (hipDeviceProp_t) */
int rt_device_info(int device, uint32_t out[4]);
const char* rt_version(void);
/* sha256 (hex) over the sources, Makefile and extra flags this library was built from (csrc/Makefile: SRC_HASH): committed
 * profiler summaries are stamped with it, so a stale binary cannot pass for the profiled one.                             */
const char* rt_source_hash(void);

#ifdef __cplusplus
}
#endif
#endif /* RT06_H */

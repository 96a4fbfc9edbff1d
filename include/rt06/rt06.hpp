// rt06.hpp — the reference's host-side C++ vocabulary, header-only, over the C ABI of rt06.h.
//
// A scene written against SuperCat908809/Ray-Tracing-v06 builds its world out of
//     Sphere / MovingSphere                      rt_engine/geometry/SphereHittable.cuh:35-52, 70-87
//     newOnDevice<LambertianAbstract<Geo>>(...)   utilities/cuda_utilities/cuda_utils.cuh:16-23, shaders/cu_materials.cuh
//     SphereHandle::MakeSphere / MakeMovingSphere rt_engine/geometry/SphereHittable.cuh:134-154
//     BVH_Handle::Factory / HittableList / bvh_node   rt_engine/geometry/BVH.cuh:64,97-100, HittableList.cuh:19, bvh_node.cuh:17
//     PinholeCamera / DefocusBlurCamera / MotionBlurCamera   rt_engine/shaders/cu_Cameras.cuh
//     Renderer::MakeRenderer / Render / DownloadRenderbuffer main/src/Renderer.h:38-46
// and this header keeps those names, argument orders and ownership rules (move-only RAII handles that own
// their objects; the Renderer borrows camera and world).  What changes is what the objects ARE: there are
// no device allocations and no vtables on the GPU — a "Hittable*" / "Material*" is a host descriptor, and
// Renderer::MakeRenderer flattens whatever world it is given (BVH, HittableList, bvh_node tree or a single
// sphere) into the three linear arrays of rt_world_flat.  Errors throw std::runtime_error (the reference's
// CUDA_ASSERT is a no-op in Release, cuError.h:25-29).
//
// Vectors: the reference's signatures use glm::vec3 / glm::vec4.  Define RT06_USE_GLM before including
// this header to use the real GLM; otherwise two layout-compatible PODs of the same names are provided.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <type_traits>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../rt06.h"

#ifdef RT06_USE_GLM
#include <glm/glm.hpp>
#else
namespace glm {
struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    float& operator[](int i) { return (&x)[i]; }
    const float& operator[](int i) const { return (&x)[i]; }
};
inline vec3 operator+(const vec3& a, const vec3& b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(const vec3& a, const vec3& b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
struct vec4 {
    float x, y, z, w;
    vec4() : x(0), y(0), z(0), w(0) {}
    vec4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
    float& operator[](int i) { return (&x)[i]; }
    const float& operator[](int i) const { return (&x)[i]; }
};
}  // namespace glm
#endif
static_assert(sizeof(glm::vec4) == 16 && sizeof(glm::vec3) == 12, "vec3/vec4 must be packed floats");

namespace rt06 {
inline void check(int rc, const char* what) {
    if (rc != RT_OK) throw std::runtime_error(std::string(what) + ": " + rt_last_error());
}
}  // namespace rt06

// ----------------------------------------------------------------------------------------------------
// aabb — rt_engine/geometry/aabb.cuh (the members scene code touches)
// ----------------------------------------------------------------------------------------------------
class aabb {
    glm::vec3 min_, max_;

public:
    aabb() : min_(1e9f), max_(-1e9f) {}
    aabb(glm::vec3 mn, glm::vec3 mx) : min_(mn), max_(mx) {}
    glm::vec3 getMin() const { return min_; }
    glm::vec3 getMax() const { return max_; }
    aabb& operator+=(const aabb& b) {
        for (int i = 0; i < 3; i++) {
            min_[i] = (b.min_[i] < min_[i]) ? b.min_[i] : min_[i];
            max_[i] = (max_[i] < b.max_[i]) ? b.max_[i] : max_[i];
        }
        return *this;
    }
};

// ----------------------------------------------------------------------------------------------------
// geometry + materials — descriptors
// ----------------------------------------------------------------------------------------------------
class Geometry {};
class Sphere : public Geometry {
public:
    glm::vec3 center;
    float radius;
    Sphere() : radius(0) {}
    Sphere(glm::vec3 c, float r) : center(c), radius(r) {}
};
class MovingSphere : public Geometry {
public:
    glm::vec3 center0, center1;
    float radius;
    MovingSphere() : radius(0) {}
    MovingSphere(glm::vec3 c0, glm::vec3 c1, float r) : center0(c0), center1(c1), radius(r) {}
};

// quad(Q,u,v) of "Ray Tracing: The Next Week" — extension, not in the reference
class Quad : public Geometry {
public:
    glm::vec3 Q, u, v;
    Quad() {}
    Quad(glm::vec3 q, glm::vec3 uu, glm::vec3 vv) : Q(q), u(uu), v(vv) {}
};

// Material (rt_engine/shaders/material.cuh:15-29) as a host descriptor
class Material {
public:
    rt_material desc{};
    virtual ~Material() = default;

protected:
    Material(uint32_t type, glm::vec3 albedo, float param, glm::vec3 albedo2 = glm::vec3(0.0f)) {
        desc.type = type;
        desc.param = param;
        for (int i = 0; i < 3; i++) { desc.albedo[i] = albedo[i]; desc.albedo2[i] = albedo2[i]; }
    }
};
template <typename G> class GeometryDependantMaterial : public Material {
protected:
    using Material::Material;
};
// GeoAcceptableMat (material.cuh:47-54) as a type trait: the material must be declared for that geometry
template <typename Geo, typename Mat> constexpr bool GeoAcceptableMat = std::is_base_of<GeometryDependantMaterial<Geo>, Mat>::value;

template <typename G> class LambertianAbstract : public GeometryDependantMaterial<G> {  // cu_materials.cuh:44-65
public:
    explicit LambertianAbstract(glm::vec3 albedo = glm::vec3(1.0f)) : GeometryDependantMaterial<G>(RT_MAT_LAMBERTIAN, albedo, 0.0f) {}
};
template <typename G> class MetalAbstract : public GeometryDependantMaterial<G> {  // cu_materials.cuh:68-96
public:
    MetalAbstract(glm::vec3 albedo, float fuzz) : GeometryDependantMaterial<G>(RT_MAT_METAL, albedo, fuzz) {}
};
template <typename G> class DielectricAbstract : public GeometryDependantMaterial<G> {  // cu_materials.cuh:106-144
public:
    DielectricAbstract(glm::vec3 albedo, float ior) : GeometryDependantMaterial<G>(RT_MAT_DIELECTRIC, albedo, ior) {}
};
template <typename G> class LambertianTexture : public GeometryDependantMaterial<G> {  // cu_materials.cuh:16-41
public:
    LambertianTexture(glm::vec3 c1, glm::vec3 c2, float scale) : GeometryDependantMaterial<G>(RT_MAT_LAMBERTIAN_CHECKER, c1, 1.0f / scale, c2) {}
};

// diffuse_light of "The Next Week" — extension, not in the reference: emits `emit`, never scatters
template <typename G> class DiffuseLightAbstract : public GeometryDependantMaterial<G> {
public:
    explicit DiffuseLightAbstract(glm::vec3 emit) : GeometryDependantMaterial<G>(RT_MAT_DIFFUSE_LIGHT, emit, 0.0f) {}
};

// isotropic / lambertian(noise_texture) / lambertian(image_texture) of "The Next Week" — extensions, not in the reference.
// A Sphere made of IsotropicAbstract is a constant_medium of that density bounded by the sphere.
template <typename G> class IsotropicAbstract : public GeometryDependantMaterial<G> {
public:
    IsotropicAbstract(glm::vec3 albedo, float density) : GeometryDependantMaterial<G>(RT_MAT_ISOTROPIC, albedo, density) {}
};
template <typename G> class NoiseTextureAbstract : public GeometryDependantMaterial<G> {   // needs Factory::SetPerlin
public:
    explicit NoiseTextureAbstract(float scale, glm::vec3 albedo = glm::vec3(0.5f)) : GeometryDependantMaterial<G>(RT_MAT_LAMBERTIAN_NOISE, albedo, scale) {}
};
template <typename G> class ImageTextureAbstract : public GeometryDependantMaterial<G> {   // needs Factory::SetImage; spheres only
public:
    ImageTextureAbstract() : GeometryDependantMaterial<G>(RT_MAT_LAMBERTIAN_IMAGE, glm::vec3(1.0f), 0.0f) {}
};

// newOnDevice<T>(args...) (cuda_utils.cuh:16-23): the reference cudaMallocs one object and runs a <<<1,1>>>
// placement-new kernel + cudaDeviceSynchronize per call; here it is a host allocation of the descriptor.
template <typename T, typename... Args> inline T* newOnDevice(const Args&... args) { return new T(args...); }

// ----------------------------------------------------------------------------------------------------
// Hittable hierarchy — host descriptors
// ----------------------------------------------------------------------------------------------------
class Hittable {
public:
    enum Kind { SPHERE, MOVING_SPHERE, QUAD, LIST, NODE, BVH_WORLD };
    virtual ~Hittable() = default;
    virtual Kind kind() const = 0;
};
class SphereHittable : public Hittable {
public:
    Sphere sphere;
    const Material* mat_ptr;
    SphereHittable(const Sphere& s, const Material* m) : sphere(s), mat_ptr(m) {}
    Kind kind() const override { return SPHERE; }
};
class MovingSphereHittable : public Hittable {
public:
    MovingSphere moving_sphere;
    const Material* mat_ptr;
    MovingSphereHittable(const MovingSphere& s, const Material* m) : moving_sphere(s), mat_ptr(m) {}
    Kind kind() const override { return MOVING_SPHERE; }
};
class QuadHittable : public Hittable {  // extension
public:
    Quad quad;
    const Material* mat_ptr;
    QuadHittable(const Quad& q, const Material* m) : quad(q), mat_ptr(m) {}
    Kind kind() const override { return QUAD; }
};
// camera::background of "The Next Week" — extension; unset = the reference's sky gradient (Renderer.cu:150-151)
struct Background {
    bool constant = false;
    glm::vec3 color{0.0f};
};
// HittableList(objects, object_count, bounds) — HittableList.cuh:19
class HittableList : public Hittable {
public:
    std::vector<const Hittable*> objects;
    aabb bounds;
    Background background;
    HittableList(const Hittable** objs, int object_count, const aabb& b) : objects(objs, objs + object_count), bounds(b) {}
    Kind kind() const override { return LIST; }
};
// bvh_node(left, right, bounds) — bvh_node.cuh:17
class bvh_node : public Hittable {
public:
    const Hittable* left;
    const Hittable* right;
    aabb bounds;
    bvh_node(const Hittable* l, const Hittable* r, const aabb& b) : left(l), right(r), bounds(b) {}
    Kind kind() const override { return NODE; }
};

namespace rt06 {
// Collects spheres and materials of a world into an rt_scene, de-duplicating shared materials.
class SceneBuilder {
    rt_scene* s_ = nullptr;
    std::unordered_map<const Material*, int32_t> mats_;

public:
    SceneBuilder() { check(rt_scene_create(&s_), "rt_scene_create"); }
    ~SceneBuilder() { rt_scene_destroy(s_); }
    SceneBuilder(const SceneBuilder&) = delete;
    SceneBuilder& operator=(const SceneBuilder&) = delete;
    rt_scene* get() const { return s_; }
    rt_scene* release() { rt_scene* s = s_; s_ = nullptr; return s; }
    int32_t material(const Material* m) {
        if (!m) throw std::runtime_error("null material");
        auto it = mats_.find(m);
        if (it != mats_.end()) return it->second;
        int32_t id = 0;
        check(rt_scene_add_material(s_, m->desc.type, m->desc.albedo, m->desc.param, m->desc.albedo2, &id), "rt_scene_add_material");
        mats_[m] = id;
        return id;
    }
    // returns the primitive index
    int32_t primitive(const Hittable* h) {
        int32_t prim = 0;
        if (h->kind() == Hittable::SPHERE) {
            auto* sh = static_cast<const SphereHittable*>(h);
            float c[3] = {sh->sphere.center[0], sh->sphere.center[1], sh->sphere.center[2]};
            check(rt_scene_add_sphere(s_, c, sh->sphere.radius, material(sh->mat_ptr), &prim), "rt_scene_add_sphere");
        } else if (h->kind() == Hittable::MOVING_SPHERE) {
            auto* mh = static_cast<const MovingSphereHittable*>(h);
            float c0[3] = {mh->moving_sphere.center0[0], mh->moving_sphere.center0[1], mh->moving_sphere.center0[2]};
            float c1[3] = {mh->moving_sphere.center1[0], mh->moving_sphere.center1[1], mh->moving_sphere.center1[2]};
            check(rt_scene_add_moving_sphere(s_, c0, c1, mh->moving_sphere.radius, material(mh->mat_ptr), &prim), "rt_scene_add_moving_sphere");
        } else if (h->kind() == Hittable::QUAD) {
            auto* qh = static_cast<const QuadHittable*>(h);
            float Q[3] = {qh->quad.Q[0], qh->quad.Q[1], qh->quad.Q[2]}, u[3] = {qh->quad.u[0], qh->quad.u[1], qh->quad.u[2]};
            float v[3] = {qh->quad.v[0], qh->quad.v[1], qh->quad.v[2]};
            check(rt_scene_add_quad(s_, Q, u, v, material(qh->mat_ptr), &prim), "rt_scene_add_quad");
        } else {
            throw std::runtime_error("only spheres and quads can be leaves of a world");
        }
        return prim;
    }
    void perlin(uint64_t seed) { check(rt_scene_set_perlin(s_, seed), "rt_scene_set_perlin"); }
    void image(uint32_t w, uint32_t h, const uint8_t* rgb) { check(rt_scene_set_image(s_, w, h, rgb), "rt_scene_set_image"); }
    void background(const Background& b) {
        float c[3] = {b.color[0], b.color[1], b.color[2]};
        check(rt_scene_set_background(s_, b.constant ? 1u : 0u, c), "rt_scene_set_background");
    }
    // child reference of a bvh_node tree: >= 0 node, < 0 primitive
    int32_t tree_ref(const Hittable* h) {
        if (h->kind() == Hittable::NODE) {
            auto* n = static_cast<const bvh_node*>(h);
            int32_t l = tree_ref(n->left), r = tree_ref(n->right), out = 0;
            float mn[3] = {n->bounds.getMin()[0], n->bounds.getMin()[1], n->bounds.getMin()[2]};
            float mx[3] = {n->bounds.getMax()[0], n->bounds.getMax()[1], n->bounds.getMax()[2]};
            check(rt_scene_add_bvh_node(s_, l, r, mn, mx, &out), "rt_scene_add_bvh_node");
            return out;
        }
        return -primitive(h) - 1;
    }
};
}  // namespace rt06

// BVH (rt_engine/geometry/BVH.cuh:13-40) — the world object BVH_Handle owns
class BVH : public Hittable {
public:
    rt_scene* scene;
    explicit BVH(rt_scene* s) : scene(s) {}
    Kind kind() const override { return BVH_WORLD; }
};

// ----------------------------------------------------------------------------------------------------
// SphereHandle — rt_engine/geometry/SphereHittable.cuh:105-158 (move-only; owns material + geometry + hittable)
// ----------------------------------------------------------------------------------------------------
class SphereHandle {
    aabb bounds;
    std::unique_ptr<Material> material_ptr;
    std::unique_ptr<Hittable> hittable_ptr;
    SphereHandle() = default;

public:
    SphereHandle(SphereHandle&&) = default;
    SphereHandle& operator=(SphereHandle&&) = default;

    template <typename MatType> static SphereHandle MakeSphere(const Sphere& sphere, MatType* mat_ptr) {
        static_assert(GeoAcceptableMat<Sphere, MatType>, "material is not declared for Sphere (material.cuh:47-54)");
        SphereHandle sp;
        glm::vec3 r(sphere.radius);
        sp.bounds = aabb(sphere.center - r, sphere.center + r);  // getSphereBounds, SphereHittable.cu:52-54
        sp.material_ptr.reset(mat_ptr);
        sp.hittable_ptr.reset(new SphereHittable(sphere, mat_ptr));
        return sp;
    }
    template <typename MatType> static SphereHandle MakeMovingSphere(const MovingSphere& sphere, MatType* mat_ptr) {
        static_assert(GeoAcceptableMat<MovingSphere, MatType>, "material is not declared for MovingSphere (material.cuh:47-54)");
        SphereHandle sp;
        glm::vec3 r(sphere.radius);
        sp.bounds = aabb(sphere.center0 - r, sphere.center0 + r);  // getMovingSphereBounds, SphereHittable.cu:85-89
        sp.bounds += aabb(sphere.center1 - r, sphere.center1 + r);
        sp.material_ptr.reset(mat_ptr);
        sp.hittable_ptr.reset(new MovingSphereHittable(sphere, mat_ptr));
        return sp;
    }
    const Hittable* getHittablePtr() const { return hittable_ptr.get(); }
    aabb getBounds() const { return bounds; }
};

// QuadHandle — extension in SphereHandle's shape (move-only; owns material + hittable).  The material may be shared
// between quads (the Cornell walls): pass owns_material = false for all but one handle.
class QuadHandle {
    aabb bounds;
    std::unique_ptr<Material> material_ptr;
    std::unique_ptr<Hittable> hittable_ptr;
    QuadHandle() = default;

public:
    QuadHandle(QuadHandle&&) = default;
    QuadHandle& operator=(QuadHandle&&) = default;
    ~QuadHandle() = default;
    template <typename MatType> static QuadHandle MakeQuad(const Quad& quad, MatType* mat_ptr, bool owns_material = true) {
        static_assert(GeoAcceptableMat<Quad, MatType>, "material is not declared for Quad");
        QuadHandle qh;
        // set_bounding_box of the book: the two diagonals' boxes; the library pads zero-thickness axes when it builds
        glm::vec3 a = quad.Q, b = quad.Q + quad.u + quad.v, c = quad.Q + quad.u, d = quad.Q + quad.v;
        auto lo = [](float x, float y) { return y < x ? y : x; };
        auto hi = [](float x, float y) { return x < y ? y : x; };
        qh.bounds = aabb(glm::vec3(lo(a[0], b[0]), lo(a[1], b[1]), lo(a[2], b[2])), glm::vec3(hi(a[0], b[0]), hi(a[1], b[1]), hi(a[2], b[2])));
        qh.bounds += aabb(glm::vec3(lo(c[0], d[0]), lo(c[1], d[1]), lo(c[2], d[2])), glm::vec3(hi(c[0], d[0]), hi(c[1], d[1]), hi(c[2], d[2])));
        if (owns_material) qh.material_ptr.reset(mat_ptr);
        qh.hittable_ptr.reset(new QuadHittable(quad, mat_ptr));
        return qh;
    }
    const Hittable* getHittablePtr() const { return hittable_ptr.get(); }
    aabb getBounds() const { return bounds; }
};

// ----------------------------------------------------------------------------------------------------
// BVH_Handle + Factory — rt_engine/geometry/BVH.cuh:42-101
// ----------------------------------------------------------------------------------------------------
class BVH_Handle {
    std::unique_ptr<BVH> bvh_;
    aabb bounds_;
    BVH_Handle(rt_scene* s, aabb b) : bvh_(new BVH(s)), bounds_(b) {}

public:
    class Factory;
    ~BVH_Handle() { if (bvh_) rt_scene_destroy(bvh_->scene); }
    BVH_Handle(BVH_Handle&&) = default;
    BVH_Handle& operator=(BVH_Handle&&) = default;
    const BVH* getBVHPtr() const { return bvh_.get(); }
    aabb getBounds() const { return bounds_; }
};

class BVH_Handle::Factory {
    std::vector<std::tuple<aabb, const Hittable*>>& arr;
    std::unique_ptr<rt06::SceneBuilder> builder_;
    Background background_;
    bool has_perlin_ = false;
    uint64_t perlin_seed_ = 0;
    uint32_t image_w_ = 0, image_h_ = 0;
    std::vector<uint8_t> image_;
    void collect() {
        builder_.reset(new rt06::SceneBuilder());
        if (has_perlin_) builder_->perlin(perlin_seed_);
        if (!image_.empty()) builder_->image(image_w_, image_h_, image_.data());
        for (auto& e : arr) builder_->primitive(std::get<1>(e));
        builder_->background(background_);
    }

public:
    explicit Factory(std::vector<std::tuple<aabb, const Hittable*>>& a) : arr(a) {}
    void SetBackground(glm::vec3 color) { background_.constant = true; background_.color = color; }  // extension; call before Build*
    void SetPerlin(uint64_t seed) { has_perlin_ = true; perlin_seed_ = seed; }                       // extension: noise tables of the world
    void SetImage(uint32_t w, uint32_t h, const uint8_t* rgb) { image_w_ = w; image_h_ = h; image_.assign(rgb, rgb + (size_t)w * h * 3); }
    void BuildBVH_TopDown() { collect(); rt06::check(rt_scene_build_bvh_topdown(builder_->get()), "BuildBVH_TopDown"); }  // _build_bvh_rec1
    void BuildBVH_TopDown_SAH() { collect(); rt06::check(rt_scene_build_bvh_sah(builder_->get()), "BuildBVH_TopDown_SAH"); }  // _build_bvh_rec2 (the #else branch, BVH.cu:168-172)
    void BuildBVH_BottomUp() { collect(); rt06::check(rt_scene_build_bvh_bottomup(builder_->get()), "BuildBVH_BottomUp"); }
    BVH_Handle* MakeHandle() {
        if (!builder_) throw std::runtime_error("BVH_Handle::Factory::MakeHandle before a Build call");
        rt_world_flat w;
        rt06::check(rt_scene_get_flat(builder_->get(), &w), "rt_scene_get_flat");
        aabb b(glm::vec3(w.bounds_min[0], w.bounds_min[1], w.bounds_min[2]), glm::vec3(w.bounds_max[0], w.bounds_max[1], w.bounds_max[2]));
        return new BVH_Handle(builder_->release(), b);
    }
};

// ----------------------------------------------------------------------------------------------------
// cameras — rt_engine/shaders/cu_Cameras.cuh (PODs; same constructor arguments)
// ----------------------------------------------------------------------------------------------------
namespace rt06 {
inline void v3(const glm::vec3& v, float out[3]) { out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; }
}
struct PinholeCamera {
    rt_camera cam{};
    PinholeCamera() = default;
    PinholeCamera(glm::vec3 lookfrom, glm::vec3 lookat, glm::vec3 up, float vfov, float aspect_ratio) {
        float a[3], b[3], c[3];
        rt06::v3(lookfrom, a); rt06::v3(lookat, b); rt06::v3(up, c);
        rt06::check(rt_camera_pinhole(a, b, c, vfov, aspect_ratio, &cam), "PinholeCamera");
    }
};
struct DefocusBlurCamera {
    rt_camera cam{};
    DefocusBlurCamera() = default;
    DefocusBlurCamera(glm::vec3 lookfrom, glm::vec3 lookat, glm::vec3 up, float vfov, float aspect_ratio, float aperture, float focus_dist) {
        float a[3], b[3], c[3];
        rt06::v3(lookfrom, a); rt06::v3(lookat, b); rt06::v3(up, c);
        rt06::check(rt_camera_defocus(a, b, c, vfov, aspect_ratio, aperture, focus_dist, &cam), "DefocusBlurCamera");
    }
};
struct MotionBlurCamera {
    rt_camera cam{};
    MotionBlurCamera() = default;
    MotionBlurCamera(glm::vec3 lookfrom, glm::vec3 lookat, glm::vec3 up, float vfov, float aspect_ratio, float time0, float time1) {
        float a[3], b[3], c[3];
        rt06::v3(lookfrom, a); rt06::v3(lookat, b); rt06::v3(up, c);
        rt06::check(rt_camera_motion(a, b, c, vfov, aspect_ratio, time0, time1, &cam), "MotionBlurCamera");
    }
};

// ----------------------------------------------------------------------------------------------------
// Renderer — main/src/Renderer.h:12-47
// ----------------------------------------------------------------------------------------------------
class Renderer {
    struct M {
        uint32_t render_width{}, render_height{};
        uint32_t samples_per_pixel{}, max_depth{};
        rt_renderer* r{};          // one GPU
        rt_multi_renderer* mr{};   // n_gpus > 1: tile shards on every GPU, one RCCL gather at frame end
    } m;
    explicit Renderer(M mm) : m(mm) {}
    Renderer(const Renderer&) = delete;
    Renderer& operator=(const Renderer&) = delete;
    void destroy() { rt_renderer_destroy(m.r); rt_multi_renderer_destroy(m.mr); m.r = nullptr; m.mr = nullptr; }

    static void flatten(const Hittable* world, rt06::SceneBuilder& tmp, rt_world_flat& w) {
        switch (world->kind()) {
        case Hittable::BVH_WORLD:
            rt06::check(rt_scene_get_flat(static_cast<const BVH*>(world)->scene, &w), "rt_scene_get_flat");
            return;
        case Hittable::LIST:
            for (const Hittable* h : static_cast<const HittableList*>(world)->objects) tmp.primitive(h);
            tmp.background(static_cast<const HittableList*>(world)->background);
            rt06::check(rt_scene_set_world_list(tmp.get()), "rt_scene_set_world_list");
            break;
        case Hittable::NODE:
            rt06::check(rt_scene_set_world_node_tree(tmp.get(), tmp.tree_ref(world)), "rt_scene_set_world_node_tree");
            break;
        default:
            tmp.primitive(world);
            rt06::check(rt_scene_set_world_list(tmp.get()), "rt_scene_set_world_list");
        }
        rt06::check(rt_scene_get_flat(tmp.get(), &w), "rt_scene_get_flat");
    }
    static Renderer make(uint32_t w_, uint32_t h_, uint32_t spp, uint32_t depth, const rt_camera& cam, const Hittable* world, uint64_t seed, int device,
                         uint32_t n_gpus, uint32_t variant) {
        if (!world) throw std::runtime_error("Renderer::MakeRenderer: null world");
        rt06::SceneBuilder tmp;
        rt_world_flat wf;
        flatten(world, tmp, wf);
        rt_render_config cfg{};
        cfg.width = w_; cfg.height = h_; cfg.samples_per_pixel = spp; cfg.max_depth = depth;
        cfg.seed = seed; cfg.device = device; cfg.rank = 0; cfg.world_size = 1; cfg.variant = variant;
        M mm;
        mm.render_width = w_; mm.render_height = h_; mm.samples_per_pixel = spp; mm.max_depth = depth;
        if (n_gpus > 1) rt06::check(rt_multi_renderer_create(&cfg, &cam, &wf, n_gpus, nullptr, &mm.mr), "Renderer::MakeRenderer");
        else rt06::check(rt_renderer_create(&cfg, &cam, &wf, &mm.r), "Renderer::MakeRenderer");
        return Renderer(mm);
    }

public:
    ~Renderer() { destroy(); }
    Renderer(Renderer&& o) : m(o.m) { o.m.r = nullptr; o.m.mr = nullptr; }
    Renderer& operator=(Renderer&& o) {
        if (this != &o) { destroy(); m = o.m; o.m.r = nullptr; o.m.mr = nullptr; }
        return *this;
    }
    // The reference takes `const MotionBlurCamera*`; the other two camera types are accepted as well.
    // seed: the reference hard-codes 1984 (Renderer.cu:51).  n_gpus > 1: the frame is tile-sharded over GPUs 0 .. n_gpus-1 of the
    // node and gathered on GPU 0 with one RCCL exchange (rt_multi_renderer_*); the image is the same for every n_gpus.
    // variant: rt_render_config::variant — 0 (default: the fastest kernel that renders the reference's bits); kToleranceMode opts a sphere world of the
    // reference's feature set into box tests by reciprocal multiplication (inside |delta| < 1e-3, not bit-exact by construction, ~1.25x faster).
    static constexpr uint32_t kToleranceMode = 6;
    static Renderer MakeRenderer(uint32_t render_width, uint32_t render_height, uint32_t samples_per_pixel, uint32_t max_depth,
                                 const MotionBlurCamera* cam, const Hittable* d_world_ptr, uint64_t seed = 1984, int device = 0, uint32_t n_gpus = 1,
                                 uint32_t variant = 0) {
        return make(render_width, render_height, samples_per_pixel, max_depth, cam->cam, d_world_ptr, seed, device, n_gpus, variant);
    }
    static Renderer MakeRenderer(uint32_t render_width, uint32_t render_height, uint32_t samples_per_pixel, uint32_t max_depth,
                                 const DefocusBlurCamera* cam, const Hittable* d_world_ptr, uint64_t seed = 1984, int device = 0, uint32_t n_gpus = 1,
                                 uint32_t variant = 0) {
        return make(render_width, render_height, samples_per_pixel, max_depth, cam->cam, d_world_ptr, seed, device, n_gpus, variant);
    }
    static Renderer MakeRenderer(uint32_t render_width, uint32_t render_height, uint32_t samples_per_pixel, uint32_t max_depth,
                                 const PinholeCamera* cam, const Hittable* d_world_ptr, uint64_t seed = 1984, int device = 0, uint32_t n_gpus = 1,
                                 uint32_t variant = 0) {
        return make(render_width, render_height, samples_per_pixel, max_depth, cam->cam, d_world_ptr, seed, device, n_gpus, variant);
    }
    void Render() {
        if (m.mr) rt06::check(rt_multi_renderer_render(m.mr), "Renderer::Render");
        else rt06::check(rt_renderer_render(m.r), "Renderer::Render");
    }
    float LastKernelMs() {
        float ms = 0;
        if (m.mr) { float t[3]; rt06::check(rt_multi_renderer_times(m.mr, t), "Renderer::LastKernelMs"); return t[0]; }
        rt06::check(rt_renderer_last_kernel_ms(m.r, &ms), "Renderer::LastKernelMs");
        return ms;
    }
    void DownloadRenderbuffer(glm::vec4* host_dst) const {
        const size_t n = (size_t)m.render_width * m.render_height * 4;
        if (m.mr) rt06::check(rt_multi_renderer_download(m.mr, reinterpret_cast<float*>(host_dst), n), "Renderer::DownloadRenderbuffer");
        else rt06::check(rt_renderer_download(m.r, reinterpret_cast<float*>(host_dst), n), "Renderer::DownloadRenderbuffer");
    }
};

// The Cornell box of "Ray Tracing: The Next Week" (BASELINE.json configs[3]) written against the reference-shaped
// C++ API (include/rt06/rt06.hpp) the way the reference's Scenes.cu writes its sphere scenes: materials through
// newOnDevice, geometry through handles, the world through BVH_Handle::Factory.  Quads, diffuse lights and the
// constant background are this build's extension (the reference has none of them).
//   cornell_app flatten                      -> prints a hash of the flat world (compared with rt_scene_cornell_box)
//   cornell_app render W H spp depth [ppm]   -> renders and prints a hash of the framebuffer
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <tuple>
#include <vector>

#include <rt06/rt06.hpp>

using LambertianQuad = LambertianAbstract<Quad>;
using LightQuad = DiffuseLightAbstract<Quad>;

class SceneCornell {
    std::vector<QuadHandle> quad_handles;
    std::unique_ptr<BVH_Handle> world_bvh;

public:
    const Hittable* getWorldPtr() const { return world_bvh->getBVHPtr(); }

    class Factory {
        std::vector<QuadHandle> handles;
        std::vector<std::tuple<aabb, const Hittable*>> bounding_boxes;

        template <typename Mat> void add(glm::vec3 Q, glm::vec3 u, glm::vec3 v, Mat* mat, bool owns) {
            handles.push_back(QuadHandle::MakeQuad(Quad(Q, u, v), mat, owns));
            bounding_boxes.push_back({handles.back().getBounds(), handles.back().getHittablePtr()});
        }
        static glm::vec3 rot_y(glm::vec3 p, float c, float s) { return glm::vec3(c * p[0] + s * p[2], p[1], -s * p[0] + c * p[2]); }
        // box(a, b, mat) of the book, then rotate_y(degrees) and translate(offset) applied to the six quads
        void box(glm::vec3 a, glm::vec3 b, float degrees, glm::vec3 offset, LambertianQuad* mat) {
            glm::vec3 mn(std::fmin(a[0], b[0]), std::fmin(a[1], b[1]), std::fmin(a[2], b[2]));
            glm::vec3 mx(std::fmax(a[0], b[0]), std::fmax(a[1], b[1]), std::fmax(a[2], b[2]));
            glm::vec3 dx(mx[0] - mn[0], 0, 0), dy(0, mx[1] - mn[1], 0), dz(0, 0, mx[2] - mn[2]);
            glm::vec3 ndx(-dx[0], -0.0f, -0.0f), ndz(-0.0f, -0.0f, -dz[2]);
            float rad = degrees * 0.01745329251994329576923690768489f, c = std::cos(rad), s = std::sin(rad);
            glm::vec3 Qs[6] = {glm::vec3(mn[0], mn[1], mx[2]), glm::vec3(mx[0], mn[1], mx[2]), glm::vec3(mx[0], mn[1], mn[2]),
                               glm::vec3(mn[0], mn[1], mn[2]), glm::vec3(mn[0], mx[1], mx[2]), glm::vec3(mn[0], mn[1], mn[2])};
            glm::vec3 us[6] = {dx, ndz, ndx, dz, dx, dx};
            glm::vec3 vs[6] = {dy, dy, dy, dy, ndz, dz};
            for (int k = 0; k < 6; k++) add(rot_y(Qs[k], c, s) + offset, rot_y(us[k], c, s), rot_y(vs[k], c, s), mat, false);
        }

    public:
        SceneCornell* MakeScene() {
            auto red = newOnDevice<LambertianQuad>(glm::vec3(0.65f, 0.05f, 0.05f));
            auto white = newOnDevice<LambertianQuad>(glm::vec3(0.73f, 0.73f, 0.73f));
            auto green = newOnDevice<LambertianQuad>(glm::vec3(0.12f, 0.45f, 0.15f));
            auto light = newOnDevice<LightQuad>(glm::vec3(15.0f, 15.0f, 15.0f));
            add(glm::vec3(555, 0, 0), glm::vec3(0, 555, 0), glm::vec3(0, 0, 555), green, true);
            add(glm::vec3(0, 0, 0), glm::vec3(0, 555, 0), glm::vec3(0, 0, 555), red, true);
            add(glm::vec3(343, 554, 332), glm::vec3(-130, 0, 0), glm::vec3(0, 0, -105), light, true);
            add(glm::vec3(0, 0, 0), glm::vec3(555, 0, 0), glm::vec3(0, 0, 555), white, true);
            add(glm::vec3(555, 555, 555), glm::vec3(-555, 0, 0), glm::vec3(0, 0, -555), white, false);
            add(glm::vec3(0, 0, 555), glm::vec3(555, 0, 0), glm::vec3(0, 555, 0), white, false);
            box(glm::vec3(0, 0, 0), glm::vec3(165, 330, 165), 15.0f, glm::vec3(265, 0, 295), white);
            box(glm::vec3(0, 0, 0), glm::vec3(165, 165, 165), -18.0f, glm::vec3(130, 0, 65), white);
            BVH_Handle::Factory bvh_factory(bounding_boxes);
            bvh_factory.SetBackground(glm::vec3(0.0f));
            bvh_factory.BuildBVH_TopDown();
            auto* scene = new SceneCornell();
            scene->world_bvh.reset(bvh_factory.MakeHandle());
            scene->quad_handles = std::move(handles);
            return scene;
        }
    };
};

// A small world through the rest of the extension vocabulary: a constant medium (a Sphere of IsotropicAbstract), Perlin marble,
// an image-textured sphere, a quad light — `cornell_app media` prints a hash that tests/test_cpp_api.py rebuilds in Python.
using IsoSphere = IsotropicAbstract<Sphere>;
using NoiseSphere = NoiseTextureAbstract<Sphere>;
using ImageSphere = ImageTextureAbstract<Sphere>;
using LambertianSphere = LambertianAbstract<Sphere>;
struct MediaScene {
    std::vector<SphereHandle> spheres;
    std::vector<QuadHandle> quads;
    std::unique_ptr<BVH_Handle> world;
};
static MediaScene* make_media_scene() {
    auto* sc = new MediaScene();
    std::vector<std::tuple<aabb, const Hittable*>> boxes;
    auto add_sphere = [&](SphereHandle&& h) { sc->spheres.push_back(std::move(h)); boxes.push_back({sc->spheres.back().getBounds(), sc->spheres.back().getHittablePtr()}); };
    add_sphere(SphereHandle::MakeSphere(Sphere(glm::vec3(0, -1000, 0), 1000.0f), newOnDevice<LambertianSphere>(glm::vec3(0.5f, 0.6f, 0.5f))));
    add_sphere(SphereHandle::MakeSphere(Sphere(glm::vec3(0, 1, 0), 1.0f), newOnDevice<IsoSphere>(glm::vec3(0.2f, 0.4f, 0.9f), 0.5f)));
    add_sphere(SphereHandle::MakeSphere(Sphere(glm::vec3(2.5f, 1, 0), 1.0f), newOnDevice<NoiseSphere>(4.0f)));
    add_sphere(SphereHandle::MakeSphere(Sphere(glm::vec3(-2.5f, 1, 0), 1.0f), newOnDevice<ImageSphere>()));
    sc->quads.push_back(QuadHandle::MakeQuad(Quad(glm::vec3(-1, 4, -1), glm::vec3(2, 0, 0), glm::vec3(0, 0, 2)), newOnDevice<LightQuad>(glm::vec3(8.0f))));
    boxes.push_back({sc->quads.back().getBounds(), sc->quads.back().getHittablePtr()});
    std::vector<uint8_t> img(8 * 4 * 3);
    for (size_t i = 0; i < img.size(); i++) img[i] = (uint8_t)(i * 7u);
    BVH_Handle::Factory f(boxes);
    f.SetBackground(glm::vec3(0.1f, 0.1f, 0.2f));
    f.SetPerlin(1984);
    f.SetImage(8, 4, img.data());
    f.BuildBVH_TopDown();
    sc->world.reset(f.MakeHandle());
    return sc;
}

static uint64_t fnv1a(const void* data, size_t n, uint64_t h = 1469598103934665603ull) {
    const unsigned char* p = static_cast<const unsigned char*>(data);
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char** argv) {
    try {
        std::string mode = argc > 1 ? argv[1] : "flatten";
        if (mode == "media") {
            MediaScene* sc = make_media_scene();
            rt_world_flat w;
            rt06::check(rt_scene_get_flat(sc->world->getBVHPtr()->scene, &w), "rt_scene_get_flat");
            uint64_t h = fnv1a(w.nodes, sizeof(rt_bvh_node) * w.n_nodes);
            for (uint32_t i = 0; i < w.n_prims; i++) {
                rt_prim pr = w.prims[i];
                rt_material m = w.materials[pr.mat & 0x7fffffffu];
                pr.mat &= 0x80000000u;
                h = fnv1a(&pr, sizeof(pr), h);
                h = fnv1a(&m, sizeof(m), h);
            }
            for (uint32_t i = 0; i < w.n_quads; i++) {
                rt_quad q = w.quads[i];
                rt_material m = w.materials[q.mat];
                q.mat = 0;
                h = fnv1a(&q, sizeof(q), h);
                h = fnv1a(&m, sizeof(m), h);
            }
            h = fnv1a(w.perlin, sizeof(rt_perlin), h);
            h = fnv1a(w.image, (size_t)w.image_width * w.image_height * 3, h);
            std::printf("media nodes=%u prims=%u quads=%u background=%u image=%ux%u fnv=%016llx\n", w.n_nodes, w.n_prims, w.n_quads, w.background,
                        w.image_width, w.image_height, (unsigned long long)h);
            delete sc;
            return 0;
        }
        SceneCornell::Factory scene_factory{};
        SceneCornell* scene_ptr = scene_factory.MakeScene();
        if (mode == "flatten") {
            rt_world_flat w;
            rt06::check(rt_scene_get_flat(static_cast<const BVH*>(scene_ptr->getWorldPtr())->scene, &w), "rt_scene_get_flat");
            uint64_t h = fnv1a(w.nodes, sizeof(rt_bvh_node) * w.n_nodes);
            for (uint32_t i = 0; i < w.n_quads; i++) {  // material ids follow first use here: hash the record, not the id
                rt_quad q = w.quads[i];
                rt_material m = w.materials[q.mat];
                q.mat = 0;
                h = fnv1a(&q, sizeof(q), h);
                h = fnv1a(&m, sizeof(m), h);
            }
            std::printf("flat nodes=%u quads=%u materials=%u root=%d max_stack=%u background=%u fnv=%016llx\n", w.n_nodes, w.n_quads,
                        w.n_materials, w.root, w.max_stack, w.background, (unsigned long long)h);
        } else {
            uint32_t width = argc > 2 ? std::atoi(argv[2]) : 600, height = argc > 3 ? std::atoi(argv[3]) : 600;
            uint32_t spp = argc > 4 ? std::atoi(argv[4]) : 16, depth = argc > 5 ? std::atoi(argv[5]) : 50;
            auto cam = new PinholeCamera(glm::vec3(278, 278, -800), glm::vec3(278, 278, 0), glm::vec3(0, 1, 0), 40.0f, width / (float)height);
            Renderer renderer = Renderer::MakeRenderer(width, height, spp, depth, cam, scene_ptr->getWorldPtr());
            std::vector<glm::vec4> fb((size_t)width * height);
            renderer.Render();
            renderer.DownloadRenderbuffer(fb.data());
            std::printf("render %ux%u spp=%u depth=%u kernel_ms=%.3f fnv=%016llx\n", width, height, spp, depth, renderer.LastKernelMs(),
                        (unsigned long long)fnv1a(fb.data(), fb.size() * sizeof(glm::vec4)));
            delete cam;
        }
        delete scene_ptr;
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}

// first_app.cpp — a caller written against the reference's API, compiled against include/rt06/rt06.hpp.
//
// It is the reference's own application flow, line for line where the API allows:
//   FirstApp::MakeApp  (main/src/FirstApp.cpp:20-56)  camera, scene factory, Renderer::MakeRenderer
//   SceneBook2BVH::Factory::_populate_world + MakeScene (rt_engine/geometry/Scenes.cu:219-315)
//   FirstApp::Run + write_renderbuffer (FirstApp.cpp:94-122)  Render, DownloadRenderbuffer, 8-bit image
// with cuHostRND (cuHostRND.h:9-32) backed by the library's host stream, so the scene it builds is the same
// one rt_scene_book2_moving() builds — tests/test_cpp_api.py checks that bit for bit.
//
//   first_app flatten                     print a checksum of the flattened world (no GPU needed)
//   first_app render W H SPP DEPTH [ppm|f32]  render, print a checksum of the float framebuffer, write a PPM (or, *.f32, the raw floats)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <tuple>
#include <vector>

#include "rt06/rt06.hpp"

// cuHostRND (cuHostRND.h:9-32): buffered uniforms, refilled in batches
class cuHostRND {
    std::vector<float> rnd_uniforms;
    size_t head = 0, base = 0;
    uint64_t seed;
    void populate() { rt06::check(rt_host_uniforms(seed, (uint32_t)base, (uint32_t)rnd_uniforms.size(), rnd_uniforms.data()), "rt_host_uniforms"); }

public:
    cuHostRND(size_t capacity, size_t seed_) : rnd_uniforms(capacity), seed(seed_) { populate(); }
    float next() {
        if (head == rnd_uniforms.size()) {
            base += rnd_uniforms.size();
            rnd_uniforms.resize(rnd_uniforms.size() * 2);
            head = 0;
            populate();
        }
        return rnd_uniforms[head++];
    }
};

// SceneBook2BVH (Scenes.h:54-92)
class SceneBook2BVH {
    BVH_Handle* bvh = nullptr;
    std::vector<SphereHandle> sphere_handles;

public:
    ~SceneBook2BVH() { delete bvh; }
    const Hittable* getWorldPtr() const { return bvh->getBVHPtr(); }

    class Factory {
        cuHostRND host_rnd{512, 1984};
        std::vector<SphereHandle> sphere_handles;

        void _populate_world() {  // Scenes.cu:219-270
            Sphere ground_sphere = Sphere(glm::vec3(0, -1000, 0), 1000.0f);
            auto ground_mat = newOnDevice<LambertianAbstract<Sphere>>(glm::vec3(0.5f));
            sphere_handles.push_back(SphereHandle::MakeSphere(ground_sphere, ground_mat));
            for (int a = -11; a < 11; a++) {
                for (int b = -11; b < 11; b++) {
#define rnd host_rnd.next()
                    float choose_mat = rnd;
                    float cx = a + rnd;  // evaluation order of the reference's `center(a + rnd, 0.2f, b + rnd)` made explicit
                    float cz = b + rnd;
                    glm::vec3 center(cx, 0.2f, cz);
                    if (choose_mat < 0.8f) {
                        float r0 = rnd, r1 = rnd, r2 = rnd, r3 = rnd, r4 = rnd, r5 = rnd;
                        auto material = newOnDevice<LambertianAbstract<MovingSphere>>(glm::vec3(r0 * r1, r2 * r3, r4 * r5));
                        float rc = rnd;
                        glm::vec3 center1 = center + glm::vec3(0, rc * 0.5f, 0);
                        sphere_handles.push_back(SphereHandle::MakeMovingSphere(MovingSphere(center, center1, 0.2f), material));
                    } else if (choose_mat < 0.95f) {
                        float r0 = rnd, r1 = rnd, r2 = rnd, r3 = rnd;
                        auto material = newOnDevice<MetalAbstract<Sphere>>(glm::vec3(0.5f * (1.0f + r0), 0.5f * (1.0f + r1), 0.5f * (1.0f + r2)), 0.5f * r3);
                        sphere_handles.push_back(SphereHandle::MakeSphere(Sphere(center, 0.2f), material));
                    } else {
                        auto material = newOnDevice<DielectricAbstract<Sphere>>(glm::vec3(1.0f), 1.5f);
                        sphere_handles.push_back(SphereHandle::MakeSphere(Sphere(center, 0.2f), material));
                    }
#undef rnd
                }
            }
            sphere_handles.push_back(SphereHandle::MakeSphere(Sphere(glm::vec3(0, 1, 0), 1), newOnDevice<DielectricAbstract<Sphere>>(glm::vec3(1.0f), 1.5f)));
            sphere_handles.push_back(SphereHandle::MakeSphere(Sphere(glm::vec3(-4, 1, 0), 1), newOnDevice<LambertianAbstract<Sphere>>(glm::vec3(0.4f, 0.2f, 0.1f))));
            sphere_handles.push_back(SphereHandle::MakeSphere(Sphere(glm::vec3(4, 1, 0), 1), newOnDevice<MetalAbstract<Sphere>>(glm::vec3(0.7f, 0.6f, 0.5f), 0.0f)));
        }

    public:
        SceneBook2BVH* MakeScene() {  // Scenes.cu:272-315
            _populate_world();
            std::vector<std::tuple<aabb, const Hittable*>> objects;
            objects.reserve(sphere_handles.size());
            for (size_t i = 0; i < sphere_handles.size(); i++)
                objects.push_back(std::make_tuple(sphere_handles[i].getBounds(), sphere_handles[i].getHittablePtr()));
            BVH_Handle::Factory bvh_factory(objects);
            bvh_factory.BuildBVH_TopDown();
            auto scene = new SceneBook2BVH();
            scene->bvh = bvh_factory.MakeHandle();
            scene->sphere_handles = std::move(sphere_handles);
            return scene;
        }
    };
};

static uint64_t fnv1a(const void* data, size_t n, uint64_t h = 1469598103934665603ull) {
    const unsigned char* p = static_cast<const unsigned char*>(data);
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

// write_renderbuffer (FirstApp.cpp:108-122): uint8 = v * 255.999f, rows flipped; PPM instead of stb's JPEG
static void write_renderbuffer(const std::string& filepath, uint32_t width, uint32_t height, const glm::vec4* data) {
    std::vector<uint8_t> img;
    img.reserve((size_t)width * height * 3);
    for (uint32_t y = 0; y < height; y++)
        for (uint32_t x = 0; x < width; x++) {
            const glm::vec4& p = data[(size_t)(height - 1 - y) * width + x];
            for (int c = 0; c < 3; c++) img.push_back(static_cast<uint8_t>(p[c] * 255.999f));
        }
    FILE* f = std::fopen(filepath.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + filepath);
    std::fprintf(f, "P6\n%u %u\n255\n", width, height);
    std::fwrite(img.data(), 1, img.size(), f);
    std::fclose(f);
}

int main(int argc, char** argv) {
    try {
        std::string mode = argc > 1 ? argv[1] : "flatten";
        SceneBook2BVH::Factory scene_factory{};
        SceneBook2BVH* scene_ptr = scene_factory.MakeScene();
        if (mode == "flatten") {
            rt_world_flat w;
            rt06::check(rt_scene_get_flat(static_cast<const BVH*>(scene_ptr->getWorldPtr())->scene, &w), "rt_scene_get_flat");
            uint64_t h = fnv1a(w.nodes, sizeof(rt_bvh_node) * w.n_nodes);
            h = fnv1a(w.prims, sizeof(rt_prim) * w.n_prims, h);
            h = fnv1a(w.materials, sizeof(rt_material) * w.n_materials, h);
            std::printf("flat nodes=%u prims=%u materials=%u root=%d max_stack=%u fnv=%016llx\n", w.n_nodes, w.n_prims, w.n_materials,
                        w.root, w.max_stack, (unsigned long long)h);
        } else {
            uint32_t width = argc > 2 ? std::atoi(argv[2]) : 1280, height = argc > 3 ? std::atoi(argv[3]) : 720;
            uint32_t spp = argc > 4 ? std::atoi(argv[4]) : 1, depth = argc > 5 ? std::atoi(argv[5]) : 4;  // FirstApp.cpp:21-39
            glm::vec3 lookfrom(13, 2, 3), lookat(0, 0, 0), up(0, 1, 0);
            float fov = 30.0f, aspect = width / (float)height;
            auto cam = new MotionBlurCamera(lookfrom, lookat, up, fov, aspect, 0.1f, 1.0f);
            // `first_app render W H spp depth out.ppm --gpus N`: tile-shard the frame over N GPUs, one RCCL gather (not in the reference)
            // `... --variant 6`: opt into the tolerance-mode kernel (Renderer::kToleranceMode; not in the reference)
            uint32_t n_gpus = 1, variant = 0;
            for (int a = 2; a + 1 < argc; a++) if (std::string(argv[a]) == "--gpus") n_gpus = (uint32_t)std::atoi(argv[a + 1]);
            for (int a = 2; a + 1 < argc; a++) if (std::string(argv[a]) == "--variant") variant = (uint32_t)std::atoi(argv[a + 1]);
            Renderer renderer = Renderer::MakeRenderer(width, height, spp, depth, cam, scene_ptr->getWorldPtr(), 1984, 0, n_gpus, variant);
            std::vector<glm::vec4> host_output_framebuffer((size_t)width * height);
            renderer.Render();
            renderer.DownloadRenderbuffer(host_output_framebuffer.data());
            std::printf("render %ux%u spp=%u depth=%u kernel_ms=%.3f fnv=%016llx\n", width, height, spp, depth, renderer.LastKernelMs(),
                        (unsigned long long)fnv1a(host_output_framebuffer.data(), host_output_framebuffer.size() * sizeof(glm::vec4)));
            if (argc > 6 && std::string(argv[6]) != "--gpus" && std::string(argv[6]) != "--variant" && std::string(argv[6]) != "-") {
                std::string path = argv[6];
                if (path.size() > 4 && path.substr(path.size() - 4) == ".f32") {   // the float framebuffer itself (row 0 = bottom), for the parity tests
                    FILE* f = std::fopen(path.c_str(), "wb");
                    if (!f) throw std::runtime_error("cannot write " + path);
                    std::fwrite(host_output_framebuffer.data(), sizeof(glm::vec4), host_output_framebuffer.size(), f);
                    std::fclose(f);
                } else write_renderbuffer(path, width, height, host_output_framebuffer.data());
            }
            delete cam;
        }
        delete scene_ptr;
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}

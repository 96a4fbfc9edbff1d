"""The reference-shaped C++ API (include/rt06/rt06.hpp): tests/cpp/first_app.cpp is the reference's own
FirstApp / SceneBook2BVH::Factory flow compiled against it.  The world it builds through
newOnDevice / SphereHandle::MakeMovingSphere / BVH_Handle::Factory must be the one rt_scene_book2_moving
builds (flat arrays bit-identical), and its rendered framebuffer must be bit-identical to the C-ABI path's."""
import os
import re
import subprocess

import numpy as np
import pytest

from _common import ROOT, pkg

APP = os.path.join(ROOT, "tests", "cpp", "first_app")


def fnv1a(chunks):
    h = 1469598103934665603
    for data in chunks:
        for b in data:
            h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def build_app():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "ray-tracing-v06_amd", "csrc")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")])


def test_cpp_scene_vocabulary_flattens_to_the_same_world():
    build_app()
    out = subprocess.check_output([APP, "flatten"], text=True)
    m = re.search(r"nodes=(\d+) prims=(\d+) materials=(\d+) root=(-?\d+) max_stack=(\d+) fnv=([0-9a-f]+)", out)
    assert m, out
    p = pkg()
    s = p.Scene.book2_moving(1984)
    nodes, prims, mats = s.arrays()
    w = s.getWorldPtr()
    assert (int(m[1]), int(m[2]), int(m[3]), int(m[4]), int(m[5])) == (w.n_nodes, w.n_prims, w.n_materials, w.root, w.max_stack)
    assert int(m[6], 16) == fnv1a([nodes.tobytes(), prims.tobytes(), mats.tobytes()])


@pytest.mark.gpu
def test_cpp_renderer_matches_c_abi_path_bit_for_bit():
    build_app()
    W, H, spp, depth = 160, 90, 6, 8
    out = subprocess.check_output([APP, "render", str(W), str(H), str(spp), str(depth)], text=True)
    m = re.search(r"fnv=([0-9a-f]+)", out)
    assert m, out
    p = pkg()
    s = p.Scene.book2_moving(1984)
    cam = p.MotionBlurCamera((13, 2, 3), (0, 0, 0), (0, 1, 0), 30.0, W / H, 0.1, 1.0)  # FirstApp.cpp:25-30
    r = p.Renderer.MakeRenderer(W, H, spp, depth, cam, s.getWorldPtr())
    r.Render()
    img = r.DownloadRenderbuffer()
    r.close()
    assert int(m[1], 16) == fnv1a([img.tobytes()])


@pytest.mark.gpu
def test_the_reference_apps_committed_frame_is_the_oracles_bit_for_bit(tmp_path):
    """The ONE render configuration the reference commits (FirstApp.cpp:21-39): the live Book-2 moving-spheres scene under its BVH, 1280x720, 1 spp,
    max_depth 4, MotionBlurCamera(lookfrom (13,2,3), lookat 0, up y, vfov 30, aspect 1280/720, t0 0.1, t1 1.0).  The full framebuffer is put
    against the CPU oracle's render of that frame — through the Python mirror (api.py -> C ABI) AND through tests/cpp/first_app, the reference's
    FirstApp flow compiled against include/rt06/rt06.hpp, with its default arguments (`render` alone = 1280 720 1 4)."""
    import _oracle as O
    from _common import bits_equal, mismatch_report
    build_app()
    W, H, spp, depth = 1280, 720, 1, 4
    oscene = O.Scene.book2_moving(1984)
    ocam = O.camera_motion((13, 2, 3), (0, 0, 0), (0, 1, 0), 30.0, W / H, 0.1, 1.0)
    ref, _ = O.render(oscene.world, ocam, W, H, spp, depth, 1984, threads=min(16, os.cpu_count() or 1))
    assert ref.shape == (H, W, 4) and np.all(ref[..., 3] == 1.0)
    # (a) Python mirror
    p = pkg()
    s = p.Scene.book2_moving(1984)
    cam = p.MotionBlurCamera((13, 2, 3), (0, 0, 0), (0, 1, 0), 30.0, W / H, 0.1, 1.0)
    r = p.Renderer.MakeRenderer(W, H, spp, depth, cam, s.getWorldPtr())
    r.Render()
    img = r.DownloadRenderbuffer()
    assert r.kernel_info()["variant"] >= 2   # the streaming kernel, not the baseline
    r.close()
    assert bits_equal(img, ref), mismatch_report(img, ref)
    # (b) the C++ mirror's app with the reference's own defaults
    raw = str(tmp_path / "frame.f32")
    out = subprocess.check_output([APP, "render", str(W), str(H), str(spp), str(depth), raw], text=True)
    m = re.search(r"render 1280x720 spp=1 depth=4 .*fnv=([0-9a-f]+)", out)
    assert m, out
    got = np.fromfile(raw, dtype=np.float32).reshape(H, W, 4)
    assert bits_equal(got, ref), mismatch_report(got, ref)
    out2 = subprocess.check_output([APP, "render"], text=True)   # no arguments: FirstApp.cpp's literals
    m2 = re.search(r"render 1280x720 spp=1 depth=4 .*fnv=([0-9a-f]+)", out2)
    assert m2 and m2[1] == m[1], out2


CORNELL = os.path.join(ROOT, "tests", "cpp", "cornell_app")


def test_cpp_quad_vocabulary_builds_the_cornell_box_prefab():
    """QuadHandle::MakeQuad / DiffuseLightAbstract / Factory::SetBackground (extension) -> the world rt_scene_cornell_box builds."""
    build_app()
    out = subprocess.check_output([CORNELL, "flatten"], text=True)
    m = re.search(r"nodes=(\d+) quads=(\d+) materials=(\d+) root=(-?\d+) max_stack=(\d+) background=(\d+) fnv=([0-9a-f]+)", out)
    assert m, out
    p = pkg()
    s = p.Scene.cornell_box()
    nodes, _, mats = s.arrays()
    quads = s.quads()
    w = s.getWorldPtr()
    assert tuple(int(m[i]) for i in range(1, 7)) == (w.n_nodes, w.n_quads, w.n_materials, w.root, w.max_stack, w.background)
    chunks = [nodes.tobytes()]
    for q in quads:
        q0 = q.copy()
        q0["mat"] = 0
        chunks += [q0.tobytes(), mats[int(q["mat"])].tobytes()]
    assert int(m[7], 16) == fnv1a(chunks)


def test_cpp_media_and_texture_vocabulary_flattens_like_the_python_one():
    """IsotropicAbstract / NoiseTextureAbstract / ImageTextureAbstract / Factory::SetPerlin / SetImage / SetBackground (C++)
    against Scene.Isotropic / NoiseTexture / ImageTexture / set_perlin / set_image / set_background (Python): same world."""
    build_app()
    out = subprocess.check_output([CORNELL, "media"], text=True)
    m = re.search(r"nodes=(\d+) prims=(\d+) quads=(\d+) background=(\d+) image=(\d+)x(\d+) fnv=([0-9a-f]+)", out)
    assert m, out
    p = pkg()
    s = p.Scene()
    s.set_perlin(1984)
    s.set_image((np.arange(8 * 4 * 3, dtype=np.uint32) * 7 % 256).astype(np.uint8).reshape(4, 8, 3))
    s.MakeSphere((0, -1000, 0), 1000.0, s.Lambertian((0.5, 0.6, 0.5)))
    s.MakeSphere((0, 1, 0), 1.0, s.Isotropic((0.2, 0.4, 0.9), 0.5))
    s.MakeSphere((2.5, 1, 0), 1.0, s.NoiseTexture(4.0))
    s.MakeSphere((-2.5, 1, 0), 1.0, s.ImageTexture())
    s.MakeQuad((-1, 4, -1), (2, 0, 0), (0, 0, 2), s.DiffuseLight((8, 8, 8)))
    s.set_background((0.1, 0.1, 0.2))
    s.BuildBVH_TopDown()
    nodes, prims, mats = s.arrays()
    quads = s.quads()
    w = s.getWorldPtr()
    assert tuple(int(m[i]) for i in range(1, 7)) == (w.n_nodes, w.n_prims, w.n_quads, w.background, w.image_width, w.image_height)
    chunks = [nodes.tobytes()]
    for pr in prims:
        p0 = pr.copy()
        p0["mat"] = pr["mat"] & 0x80000000
        chunks += [p0.tobytes(), mats[int(pr["mat"]) & 0x7fffffff].tobytes()]
    for q in quads:
        q0 = q.copy()
        q0["mat"] = 0
        chunks += [q0.tobytes(), mats[int(q["mat"])].tobytes()]
    chunks += [s.perlin_bytes(), s.image().tobytes()]
    assert int(m[7], 16) == fnv1a(chunks)


@pytest.mark.gpu
def test_cpp_cornell_render_matches_c_abi_path_bit_for_bit():
    build_app()
    W, H, spp, depth = 120, 120, 6, 50
    out = subprocess.check_output([CORNELL, "render", str(W), str(H), str(spp), str(depth)], text=True)
    m = re.search(r"fnv=([0-9a-f]+)", out)
    assert m, out
    p = pkg()
    s = p.Scene.cornell_box()
    cam = p.PinholeCamera((278, 278, -800), (278, 278, 0), (0, 1, 0), 40.0, W / H)
    r = p.Renderer.MakeRenderer(W, H, spp, depth, cam, s.getWorldPtr())
    r.Render()
    img = r.DownloadRenderbuffer()
    r.close()
    assert int(m[1], 16) == fnv1a([img.tobytes()])


def test_header_is_plain_c_and_the_library_links_from_c():
    build_app()
    out = subprocess.check_output([os.path.join(ROOT, "tests", "cpp", "abi_c_check")], text=True)
    assert "C ABI ok" in out


REF_GLM = "/root/reference/Libraries/include"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF_GLM, "glm")), reason="build container only: needs the reference's vendored GLM (absent on the GPU box)")
def test_cpp_api_compiles_against_the_references_own_glm_and_flattens_to_the_same_world(tmp_path):
    """INTEGRATION.md §2 tells a maintainer to build with -DRT06_USE_GLM so that glm::vec3 IS the reference's vendored GLM 0.9.9.7:
    the FirstApp caller must compile that way and build byte-identical flat arrays."""
    build_app()
    exe = str(tmp_path / "first_app_glm")
    libdir = os.path.join(ROOT, "ray-tracing-v06_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wno-volatile", "-Wno-deprecated-volatile", "-DRT06_USE_GLM", "-DGLM_ENABLE_EXPERIMENTAL",
                           f"-I{REF_GLM}", f"-I{os.path.join(ROOT, 'include')}", "-o", exe, os.path.join(ROOT, "tests", "cpp", "first_app.cpp"),
                           f"-L{libdir}", "-lrt06", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    assert subprocess.check_output([exe, "flatten"], text=True) == subprocess.check_output([APP, "flatten"], text=True)

"""GPU parity tests for the first widening step (BASELINE.json configs[3]): quads,
diffuse lights and a constant background — the Cornell box of "The Next Week".

None of this exists in the reference (it has spheres, three scattering materials and the sky gradient only),
so there is nothing of the reference's to pin it on: PARITY UNPINNED — the HIP path is checked against this
build's CPU oracle (oracle/rt_oracle.c: quad_closest_intersection, sample_world with accum_radiance), which
restates the book's published algorithm in the reference's arithmetic conventions.
"""
import ctypes as C

import numpy as np
import pytest

import _oracle as O
from _common import (as_oracle_camera, as_oracle_world, bits_equal, config_cameras, config_scene, mismatch_report, pkg,
                     random_mixed_scene, random_rays)

pytestmark = pytest.mark.gpu

TOL_MEASURED = 1e-5  # baseline kernel: wave-order summation, same bound as test_gpu_parity.py


@pytest.fixture(scope="module")
def p():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return pkg()


def _render_both(p, scene, cam, W, H, spp, depth=50, variant=0, seed=1984):
    w = scene.getWorldPtr()
    r = p.Renderer.MakeRenderer(W, H, spp, depth, cam, w, seed=seed, variant=variant)
    r.Render()
    img = r.DownloadRenderbuffer()
    r.close()
    ref, _ = O.render(as_oracle_world(w), as_oracle_camera(cam), W, H, spp, depth, seed)
    return img, ref


@pytest.mark.parametrize("builder", [0, 1, 2, 3])
def test_closest_intersection_over_spheres_and_quads(p, builder):
    """quad::hit next to Sphere::ClosestIntersection under every world kind: hit, t, unified primitive index, normal."""
    rng = np.random.default_rng(70 + builder)
    s, _ = random_mixed_scene(p, rng, 30, 40, builder)
    w = s.getWorldPtr()
    n = 16384
    rays = random_rays(rng, n, spread=7.0)
    rays[: n // 2, 3:6] = -rays[: n // 2, 0:3] + rng.standard_normal((n // 2, 3)).astype(np.float32)  # towards the middle
    for k in range(0, 256):
        rays[k, 3 + rng.integers(0, 3)] = 0.0   # rays parallel to axis-aligned quads: |denom| < 1e-8 branch
    hit, t, prim, nrm = p.api.probe_trace(w, rays)
    ow = as_oracle_world(w)
    ehit = np.zeros(n, np.int32); et = np.zeros(n, np.float32); eprim = np.zeros(n, np.int32); en = np.zeros((n, 3), np.float32)
    assert O.lib().orc_trace_batch(C.byref(ow), n, rays, ehit, et, eprim, en) == 0
    assert np.array_equal(hit, ehit) and np.array_equal(prim, eprim), f"{(prim != eprim).sum()} primitive indices differ"
    assert bits_equal(t, et), mismatch_report(t, et)
    assert bits_equal(nrm, en), mismatch_report(nrm, en)
    assert (prim[hit != 0] >= w.n_prims).mean() > 0.1 and (prim[hit != 0] < w.n_prims).mean() > 0.1


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
@pytest.mark.parametrize("W,H,spp", [(200, 200, 24), (77, 53, 67)])
def test_cornell_box_framebuffer_matches_oracle(p, variant, W, H, spp):
    s = config_scene(p, "cornell_box")
    cam = config_cameras(p, "cornell_box", W, H)
    img, ref = _render_both(p, s, cam, W, H, spp, variant=variant)
    assert np.all(img[..., 3] == 1.0)
    assert np.nanmax(np.abs(img - ref)) <= TOL_MEASURED * max(1.0, float(np.nanmax(ref)))
    if variant != 1:
        assert bits_equal(img, ref), mismatch_report(img, ref)
    # it is a Cornell box: the light is visible (15 clamps to 1 per sample, Renderer.cu:209-211), most of the rest is dim
    assert ref[..., :3].max() == 1.0 and 0.02 < np.nanmean(ref[..., :3]) < 0.9


def test_cornell_box_per_sample_radiance_bit_exact(p):
    W = H = 600
    s = config_scene(p, "cornell_box")
    cam = config_cameras(p, "cornell_box", W, H)
    w = s.getWorldPtr()
    rng = np.random.default_rng(8)
    n = 4096
    keys = np.stack([rng.integers(0, W * H, n), rng.integers(0, 5000, n)], axis=1).astype(np.uint32)
    cfg = p.capi.RenderConfig(W, H, 1, 50, 1984, 0, 0, 1, 0)
    got = p.api.probe_radiance(cfg, cam, w, keys)
    exp = np.zeros((n, 3), np.float32)
    ow, oc = as_oracle_world(w), as_oracle_camera(cam)
    assert O.lib().orc_radiance_batch(C.byref(ow), C.byref(oc), W, H, 50, 1984, n, keys, exp) == 0
    assert bits_equal(got, exp), mismatch_report(got, exp)
    assert 0.02 < (got > 0).any(axis=1).mean() < 0.5  # small light, open front: few paths find it


def test_filtered_variant_refuses_extended_worlds(p):
    s = config_scene(p, "cornell_box")
    cam = config_cameras(p, "cornell_box", 64, 64)
    with pytest.raises(p.capi.RtError, match="variant"):
        p.Renderer.MakeRenderer(64, 64, 1, 5, cam, s.getWorldPtr(), variant=4)


def test_max_depth_cuts_emission_like_the_oracle(p):
    """depth 1 sees only directly visible lights; each further bounce may add light, never remove it."""
    s = config_scene(p, "cornell_box")
    cam = config_cameras(p, "cornell_box", 96, 96)
    prev = None
    for depth in (1, 2, 3, 8):
        img, ref = _render_both(p, s, cam, 96, 96, 8, depth=depth)
        assert bits_equal(img, ref), f"depth {depth}: " + mismatch_report(img, ref)
        if depth == 1:
            assert np.all(img[..., :3] * 8 == np.round(img[..., :3] * 8))  # per sample: light seen (clamped to 1) or nothing
        if prev is not None:
            assert img[..., :3].sum() >= prev
        prev = img[..., :3].sum()


@pytest.mark.parametrize("seed", list(range(10)))
def test_random_sphere_quad_light_scenes_render_bit_exact(p, seed):
    """Differential fuzz over the extension: spheres (static/moving) + quads (axis-aligned and skew), all five
    material kinds incl. lights on spheres, sky or constant background, every builder, three camera types."""
    rng = np.random.default_rng(5000 + seed)
    builder = int(rng.integers(0, 4))
    bg = None if rng.random() < 0.4 else tuple(float(x) for x in rng.random(3) * 0.3)
    ns, nq = int(rng.integers(0, 30)), int(rng.integers(1, 30))
    s, _ = random_mixed_scene(p, rng, ns, nq, builder, background=bg)
    W, H = int(rng.integers(17, 97)), int(rng.integers(9, 65))
    spp, depth = int(rng.integers(1, 24)), int(rng.choice([1, 2, 5, 50]))
    eye = ((rng.random(3) * 2 - 1) * np.array([9, 4, 9])).astype(np.float32)
    cam_kind = int(rng.integers(0, 3))
    if cam_kind == 0:
        cam = p.PinholeCamera(eye, (0, 0, 0), (0, 1, 0), float(rng.uniform(20, 100)), W / H)
    elif cam_kind == 1:
        cam = p.DefocusBlurCamera(eye, (0, 0, 0), (0, 1, 0), float(rng.uniform(20, 100)), W / H, float(rng.uniform(0, 0.5)), float(rng.uniform(2, 12)))
    else:
        cam = p.MotionBlurCamera(eye, (0, 0, 0), (0, 1, 0), float(rng.uniform(20, 100)), W / H, 0.0, 1.0)
    img, ref = _render_both(p, s, cam, W, H, spp, depth=depth)
    tag = f"seed {seed} builder {builder} bg {bg} cam {cam_kind} {W}x{H}x{spp} depth {depth} spheres {ns} quads {nq}: "
    assert np.array_equal(np.isnan(img), np.isnan(ref)), tag
    assert bits_equal(img, ref), tag + mismatch_report(img, ref)   # HittableList worlds with quads run on the streaming kernel too


def test_cornell_box_sharded_over_four_ranks_is_the_same_image(p):
    """configs[3] is quoted tile-sharded on 4 GPUs: the assembled image must not depend on the rank count."""
    import torch
    W = H = 150
    spp = 12
    s = config_scene(p, "cornell_box")
    cam = config_cameras(p, "cornell_box", W, H)
    w = s.getWorldPtr()
    single = p.Renderer.MakeRenderer(W, H, spp, 50, cam, w)
    single.Render()
    ref = single.DownloadRenderbuffer()
    shards = []
    for rank in range(4):
        r = p.Renderer.MakeRenderer(W, H, spp, 50, cam, w, rank=rank, world_size=4)
        buf = torch.zeros(r.shard_floats(), dtype=torch.float32, device="cuda:0")
        r.render_async(torch.cuda.current_stream().cuda_stream, buf.data_ptr())
        torch.cuda.synchronize()
        shards.append(buf)
        last = r
    image = torch.empty(H * W * 4, dtype=torch.float32, device="cuda:0")
    last.assemble(torch.cat(shards).data_ptr(), image.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert image.cpu().numpy().reshape(H, W, 4).tobytes() == ref.tobytes()


def _math_inputs():
    rng = np.random.default_rng(31)
    u = ((rng.integers(0, 1 << 24, 1 << 16) + 1) * 2.0 ** -24).astype(np.float32)      # every uniform the RNG can produce is of this form
    xs = np.concatenate([(rng.random(1 << 16) * 4000 - 2000), [0.0, -0.0, 1e-8, 3.14159274, 1.57079637]]).astype(np.float32)
    xc = np.concatenate([(rng.random(1 << 16) * 2 - 1), [-1.0, 1.0, 0.0, 0.5, -0.5]]).astype(np.float32)
    ya = np.concatenate([rng.standard_normal(1 << 16), [0.0, 0.0, 1.0, -1.0, 0.0]]).astype(np.float32)
    xa = np.concatenate([rng.standard_normal(1 << 16), [1.0, -1.0, 0.0, 0.0, 0.0]]).astype(np.float32)
    return u, xs, xc, ya, xa


def test_extension_math_is_bit_identical_on_cpu_and_gpu(p):
    """log / sin / acos / atan2 of the extension materials: own fp32 routines, same bits in the oracle and on the GPU."""
    u, xs, xc, ya, xa = _math_inputs()
    for fn, a, b in ((0, u, u), (1, xs, xs), (2, xc, xc), (3, ya, xa)):
        exp = np.zeros_like(a)
        O.lib().orc_math_batch(fn, len(a), a, b, exp)
        got = p.api.probe_math(fn, a, b)
        assert bits_equal(got, exp), f"fn {fn}: " + mismatch_report(got, exp)


# ------------------------------------------------------------------------------------------------
# constant media (constant_medium + isotropic of "The Next Week"): the second widening step, parity unpinned as above
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("builder", [0, 1, 2, 3])
def test_closest_intersection_through_constant_media(p, builder):
    """A medium test draws a uniform (free path ~ -log(u) / density) in traversal order: hit, t, primitive must match."""
    rng = np.random.default_rng(170 + builder)
    s, _ = random_mixed_scene(p, rng, 40, 20, builder, media=True)
    w = s.getWorldPtr()
    n = 16384
    rays = random_rays(rng, n, spread=7.0)
    rays[: n // 2, 3:6] = -rays[: n // 2, 0:3] + rng.standard_normal((n // 2, 3)).astype(np.float32)
    hit, t, prim, nrm = p.api.probe_trace(w, rays)
    ow = as_oracle_world(w)
    ehit = np.zeros(n, np.int32); et = np.zeros(n, np.float32); eprim = np.zeros(n, np.int32); en = np.zeros((n, 3), np.float32)
    assert O.lib().orc_trace_batch(C.byref(ow), n, rays, ehit, et, eprim, en) == 0
    assert np.array_equal(hit, ehit) and np.array_equal(prim, eprim), f"{(prim != eprim).sum()} primitive indices differ"
    assert bits_equal(t, et), mismatch_report(t, et)
    assert bits_equal(nrm, en), mismatch_report(nrm, en)
    mats = s.arrays()[2]
    prims = s.arrays()[1]
    is_medium = mats["type"][prims["mat"] & 0x7fffffff] == 5
    hit_medium = is_medium[np.clip(prim, 0, len(prims) - 1)] & (hit != 0) & (prim < len(prims))
    assert hit_medium.mean() > 0.02   # the media do stop rays


@pytest.mark.parametrize("seed", list(range(8)))
def test_random_scenes_with_media_render_bit_exact(p, seed):
    rng = np.random.default_rng(7000 + seed)
    builder = int(rng.integers(0, 4))
    bg = None if rng.random() < 0.5 else tuple(float(x) for x in rng.random(3) * 0.3)
    ns, nq = int(rng.integers(4, 40)), int(rng.integers(0, 20))
    s, _ = random_mixed_scene(p, rng, ns, nq, builder, background=bg, media=True)
    W, H = int(rng.integers(17, 97)), int(rng.integers(9, 65))
    spp, depth = int(rng.integers(1, 24)), int(rng.choice([1, 2, 5, 50]))
    eye = ((rng.random(3) * 2 - 1) * np.array([9, 4, 9])).astype(np.float32)
    cam = p.MotionBlurCamera(eye, (0, 0, 0), (0, 1, 0), float(rng.uniform(20, 100)), W / H, 0.0, 1.0)
    img, ref = _render_both(p, s, cam, W, H, spp, depth=depth)
    tag = f"seed {seed} builder {builder} bg {bg} {W}x{H}x{spp} depth {depth} spheres {ns} quads {nq}: "
    assert np.array_equal(np.isnan(img), np.isnan(ref)), tag
    assert bits_equal(img, ref), tag + mismatch_report(img, ref)


def test_camera_inside_a_global_fog(p):
    """The Book-2 final scene wraps everything in a thin constant medium (radius 5000, density 1e-4): the camera is INSIDE it."""
    s = p.Scene()
    ground = s.Lambertian((0.48, 0.83, 0.53))
    s.MakeSphere((0, -1000, 0), 1000.0, ground)
    s.MakeSphere((0, 1, 0), 1.0, s.Dielectric((1, 1, 1), 1.5))
    s.MakeConstantMedium((0, 1, 0), 1.0, 0.2, (0.2, 0.4, 0.9))          # subsurface: a medium inside the glass ball
    s.MakeSphere((2.5, 1, 0.5), 1.0, s.Metal((0.8, 0.8, 0.9), 0.5))
    s.MakeQuad((-2, 4, -2), (3, 0, 0), (0, 0, 3), s.DiffuseLight((7, 7, 7)))
    s.MakeConstantMedium((0, 0, 0), 5000.0, 0.0001, (1, 1, 1))
    s.set_background((0, 0, 0))
    s.BuildBVH_TopDown()
    W, H, spp = 120, 80, 16
    cam = p.PinholeCamera((6, 2.5, 7), (0, 1, 0), (0, 1, 0), 35.0, W / H)
    for variant in (0, 1, 2):
        img, ref = _render_both(p, s, cam, W, H, spp, variant=variant)
        if variant == 1:
            assert np.nanmax(np.abs(img - ref)) <= TOL_MEASURED
        else:
            assert bits_equal(img, ref), f"variant {variant}: " + mismatch_report(img, ref)
    assert ref[..., :3].max() > 0.5 and ref[..., :3].mean() > 0.01


# ------------------------------------------------------------------------------------------------
# Perlin noise (marble) and image textures
# ------------------------------------------------------------------------------------------------
def _test_image(h=48, w=96):
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w, 3), np.uint8)
    img[..., 0] = (xx * 255 // (w - 1)).astype(np.uint8)
    img[..., 1] = (yy * 255 // (h - 1)).astype(np.uint8)
    img[..., 2] = (((xx // 8) + (yy // 8)) % 2 * 200 + 30).astype(np.uint8)
    return img


def test_noise_and_image_textures_render_bit_exact(p):
    """lambertian(noise_texture) on a sphere, the ground and a quad, lambertian(image_texture) on spheres (one of them
    moving): every variant against the oracle."""
    s = p.Scene()
    s.set_perlin(1984).set_image(_test_image())
    marble = s.NoiseTexture(4.0)
    marble2 = s.NoiseTexture(0.7, (0.4, 0.5, 0.3))
    earth = s.ImageTexture()
    s.MakeSphere((0, -1000, 0), 1000.0, marble2)
    s.MakeSphere((0, 2, 0), 2.0, marble)
    s.MakeSphere((4, 1.2, 1.5), 1.2, earth)
    s.MakeMovingSphere((-3.5, 1, 2.5), (-3.5, 1.6, 2.5), 1.0, earth)
    s.MakeQuad((-6, 0.5, -4), (5, 0, 0), (0, 4, 0), marble)
    s.MakeQuad((3, 5, -2), (2, 0, 0), (0, 0, 2), s.DiffuseLight((6, 6, 6)))
    s.set_background((0.35, 0.4, 0.5))
    s.BuildBVH_TopDown()
    W, H, spp = 150, 100, 12
    cam = p.MotionBlurCamera((10, 4, 9), (0, 1.5, 0), (0, 1, 0), 35.0, W / H, 0.0, 1.0)
    for variant in (0, 1, 2):
        img, ref = _render_both(p, s, cam, W, H, spp, variant=variant)
        if variant == 1:
            assert np.nanmax(np.abs(img - ref)) <= TOL_MEASURED
        else:
            assert bits_equal(img, ref), f"variant {variant}: " + mismatch_report(img, ref)
    # the textures are visible: the image sphere shows a wide range of colours
    assert ref[..., 0].std() > 0.05 and ref[..., 1].std() > 0.05


def test_textured_materials_need_their_tables(p):
    s = p.Scene()
    s.MakeSphere((0, 0, 0), 1.0, s.NoiseTexture(2.0))
    s.BuildBVH_TopDown()
    cam = p.PinholeCamera((0, 0, 5), (0, 0, 0), (0, 1, 0), 40.0, 1.0)
    with pytest.raises(p.capi.RtError, match="Perlin"):
        p.Renderer.MakeRenderer(16, 16, 1, 5, cam, s.getWorldPtr())
    s2 = p.Scene()
    s2.MakeSphere((0, 0, 0), 1.0, s2.ImageTexture())
    s2.BuildBVH_TopDown()
    with pytest.raises(p.capi.RtError, match="image"):
        p.Renderer.MakeRenderer(16, 16, 1, 5, cam, s2.getWorldPtr())


@pytest.mark.parametrize("world", ["bvh", "list"])
def test_image_texture_on_quads_renders_bit_exact(p, world):
    """image_texture on a quad: (u, v) are the planar coordinates (alpha, beta) of quad::hit ("The Next Week"), row 0 of the image at
    v = 1.  A picture on a wall, a textured floor and a tilted panel, next to an image-textured sphere; BVH and HittableList worlds,
    every kernel variant, against the oracle."""
    s = p.Scene()
    s.set_image(_test_image())
    pic = s.ImageTexture()
    s.MakeQuad((-3, 0, -2), (6, 0, 0), (0, 4, 0), pic)                 # wall
    s.MakeQuad((-4, 0, 4), (8, 0, 0), (0, 0, -6), pic)                # floor
    s.MakeQuad((2.5, 0.2, 1.5), (1.5, 0.3, 0.8), (-0.4, 1.6, 0.3), pic)   # tilted panel
    s.MakeSphere((-2, 1, 1.5), 1.0, pic)
    s.MakeSphere((0.5, 0.6, 2.5), 0.6, s.Metal((0.9, 0.9, 0.9), 0.0))
    s.set_background((0.6, 0.7, 0.9))
    (s.BuildBVH_TopDown if world == "bvh" else s.MakeHittableList)()
    W, H, spp = 150, 100, 10
    cam = p.PinholeCamera((1, 3, 9), (0, 1.2, 0), (0, 1, 0), 40.0, W / H)
    for variant in (0, 1, 2):
        img, ref = _render_both(p, s, cam, W, H, spp, variant=variant)
        if variant == 1:
            assert np.nanmax(np.abs(img - ref)) <= TOL_MEASURED
        else:
            assert bits_equal(img, ref), f"variant {variant}: " + mismatch_report(img, ref)
    assert ref[..., 0].std() > 0.05 and ref[..., 2].std() > 0.05      # the picture is visible


def test_book2_final_scene_renders_bit_exact(p):
    """final_scene() of "The Next Week" (BASELINE configs[4]): too large for the LDS -> global-memory streaming kernel."""
    W, H, spp, depth = 96, 96, 4, 40
    s = config_scene(p, "book2_final")
    cam = config_cameras(p, "book2_final", W, H)
    w = s.getWorldPtr()
    r = p.Renderer.MakeRenderer(W, H, spp, depth, cam, w)
    assert r.kernel_info()["variant"] == 3 and not r.kernel_info()["lds_resident"]
    r.Render()
    img = r.DownloadRenderbuffer()
    r.close()
    ref, _ = O.render(as_oracle_world(w), as_oracle_camera(cam), W, H, spp, depth)
    assert bits_equal(img, ref), mismatch_report(img, ref)
    assert ref[..., :3].max() == 1.0 and ref[..., :3].mean() > 0.02


# ------------------------------------------------------------------------------------------------
# BASELINE configs[3] / [4] at (near) full size through a sparse exact check: the oracle renders only the sampled pixels
# ------------------------------------------------------------------------------------------------
def _sparse_full_size(p, which, W, H, spp, depth, n_px, seed):
    s = config_scene(p, which)
    cam = config_cameras(p, which, W, H)
    w = s.getWorldPtr()
    r = p.Renderer.MakeRenderer(W, H, spp, depth, cam, w)
    r.Render()
    img = r.DownloadRenderbuffer()
    ms = r.last_kernel_ms()
    r.close()
    rng = np.random.default_rng(seed)
    gids = rng.integers(0, W * H, n_px).astype(np.uint32)
    exp = O.render_pixels(as_oracle_world(w), as_oracle_camera(cam), W, H, spp, depth, gids)
    got = img.reshape(-1, 4)[gids]
    assert np.all(img[..., 3] == 1.0) and np.all(np.isfinite(img))
    assert bits_equal(got, exp), mismatch_report(got, exp)
    print(f"{which} {W}x{H}x{spp}: {W * H * spp / ms / 1e3:.1f} Msamples/s ({ms:.1f} ms), {n_px} pixels bit-identical to the oracle")


def test_full_size_cornell_box_sparse_parity(p):
    """configs[3]: 600x600, 5000 spp (1.8e9 samples on one GPU; the config shards it over 4)."""
    _sparse_full_size(p, "cornell_box", 600, 600, 5000, 50, 24, 13)


def test_book2_final_at_bench_size_sparse_parity(p):
    """configs[4] scene at the size bench.py --workload book2_final times (800x800, 200 spp, depth 40)."""
    _sparse_full_size(p, "book2_final", 800, 800, 200, 40, 48, 14)


def test_cornell_box_as_a_hittable_list_like_the_book(p):
    """The book renders the Cornell box from a plain hittable_list: same quads under MakeHittableList must take the streaming
    kernel (not the baseline) and give the SAME image as the BVH world — every quad is tested either way, in list order."""
    s = p.Scene()
    red, white, green = s.Lambertian((0.65, 0.05, 0.05)), s.Lambertian((0.73, 0.73, 0.73)), s.Lambertian((0.12, 0.45, 0.15))
    light = s.DiffuseLight((15, 15, 15))
    s.MakeQuad((555, 0, 0), (0, 555, 0), (0, 0, 555), green)
    s.MakeQuad((0, 0, 0), (0, 555, 0), (0, 0, 555), red)
    s.MakeQuad((343, 554, 332), (-130, 0, 0), (0, 0, -105), light)
    s.MakeQuad((0, 0, 0), (555, 0, 0), (0, 0, 555), white)
    s.MakeQuad((555, 555, 555), (-555, 0, 0), (0, 0, -555), white)
    s.MakeQuad((0, 0, 555), (555, 0, 0), (0, 555, 0), white)
    s.MakeBox((0, 0, 0), (165, 330, 165), white, 15.0, (265, 0, 295))
    s.MakeBox((0, 0, 0), (165, 165, 165), white, -18.0, (130, 0, 65))
    s.set_background((0, 0, 0))
    s.MakeHittableList()
    W = H = 120
    cam = config_cameras(p, "cornell_box", W, H)
    r = p.Renderer.MakeRenderer(W, H, 16, 50, cam, s.getWorldPtr())
    assert r.kernel_info()["variant"] == 2 and r.kernel_info()["lds_resident"]
    r.close()
    img, ref = _render_both(p, s, cam, W, H, 16)
    assert bits_equal(img, ref), mismatch_report(img, ref)
    bvh = config_scene(p, "cornell_box")
    img_bvh, _ = _render_both(p, bvh, cam, W, H, 16)
    assert img_bvh.tobytes() == img.tobytes()

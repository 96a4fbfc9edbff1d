"""GPU parity tests (-m gpu): every hot-path function on the MI355X, called through the C ABI
(librt06.so), against the CPU oracle on the same seeded inputs.

Bars:  * integer / index results (hit flags, primitive indices, draw counts, sphere-index map): exact.
       * per-function float results and per-sample radiance: BIT-exact (the arithmetic contract of
         DESIGN.md makes CPU and GPU evaluate the same IEEE expression tree).
       * framebuffer: per-channel |delta| < 1e-3 (BASELINE.json north_star); the only source of
         difference is the summation order of the per-pixel mean, so the tests also assert a much
         tighter measured bound (1e-5).
"""
import ctypes as C

import numpy as np
import pytest

import _oracle as O
from _common import (as_oracle_camera, as_oracle_world, bits_equal, config_cameras, config_scene,
                     mismatch_report, oracle_scene, pkg, random_rays)

pytestmark = pytest.mark.gpu

TOL_SPEC = 1e-3       # BASELINE.json: per-channel |delta| < 1e-3 vs the reference image
TOL_MEASURED = 1e-5   # what reassociating the per-pixel float sum can cost (post-gamma, [0,1] values)


@pytest.fixture(scope="module")
def p():
    m = pkg()
    assert m.api.device_count() >= 1, "no HIP device: the gpu tests need an MI355X"
    return m


SPECIALS = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-9, -1e-9, 1e-38, 1e-42, 3.4028235e38,
                     -3.4028235e38, 0.5, 2.0, 1e9, -1e9], dtype=np.float32)


# ------------------------------------------------------------------------------------------------
def test_rng_stream_bit_exact(p):
    rng = np.random.default_rng(0)
    keys = rng.integers(0, 2**32, size=(512, 2), dtype=np.uint32)
    got = p.api.probe_rng(1984, keys, 37)
    exp = np.zeros_like(got)
    for i, (px, s) in enumerate(keys):
        O.lib().orc_rng_uniforms(1984, int(px), int(s), 0, 37, exp[i])
    assert bits_equal(got, exp), mismatch_report(got, exp)


def test_aabb_intersects_bit_exact(p):
    """G1: aabb::intersects incl. axis-parallel rays (d_i = 0), origin inside, degenerate boxes, inf/NaN."""
    rng = np.random.default_rng(1)
    n = 8192
    lo = (rng.random((n, 3), dtype=np.float32) * 2 - 1) * 10
    ext = rng.random((n, 3), dtype=np.float32) * 4
    boxes = np.concatenate([lo, lo + ext], axis=1).astype(np.float32)
    rays = random_rays(rng, n, with_time=False)
    maxd = np.where(rng.random(n) < 0.5, np.float32(3.402823466e38), rng.random(n, dtype=np.float32) * 30).astype(np.float32)
    # axis-parallel directions
    for k in range(0, 1500):
        rays[k, 3 + rng.integers(0, 3)] = 0.0 if k % 2 else -0.0
    # origins inside the box
    rays[1500:2500, 0:3] = lo[1500:2500] + ext[1500:2500] * rng.random((1000, 3), dtype=np.float32)
    # origins exactly on a face with a zero direction component -> 0/0
    rays[2500:2700, 0] = boxes[2500:2700, 0]
    rays[2500:2700, 3] = 0.0
    # degenerate (flat / inverted / empty aabb() = (1e9,-1e9)) boxes
    boxes[2700:2900, 3:6] = boxes[2700:2900, 0:3]
    boxes[2900:3000, 0:3] = 1e9
    boxes[2900:3000, 3:6] = -1e9
    # special values sprinkled everywhere
    for k in range(3000, 3600):
        rays[k, rng.integers(0, 6)] = SPECIALS[rng.integers(0, len(SPECIALS))]
        boxes[k, rng.integers(0, 6)] = SPECIALS[rng.integers(0, len(SPECIALS))]
    hit, dist = p.api.probe_aabb(boxes, rays, maxd)
    ehit = np.zeros(n, np.int32); edist = np.zeros(n, np.float32)
    O.lib().orc_aabb_batch(n, boxes, rays, maxd, ehit, edist)
    assert np.array_equal(hit, ehit), f"{(hit != ehit).sum()} hit flags differ"
    assert bits_equal(dist, edist), mismatch_report(dist, edist)
    assert 0.05 < hit.mean() < 0.95


def test_sphere_closest_intersection_bit_exact(p):
    """G2: _sphere_closest_intersection incl. tangent, origin-inside, behind, un-normalised d."""
    rng = np.random.default_rng(2)
    n = 8192
    spheres = np.concatenate([(rng.random((n, 3), dtype=np.float32) * 2 - 1) * 8,
                              rng.random((n, 1), dtype=np.float32) * 2 + 0.05], axis=1).astype(np.float32)
    rays = random_rays(rng, n, with_time=False)
    # aim most rays at their sphere (with an offset up to 1.5 radii -> hits, grazes and misses)
    k = np.arange(0, 6000)
    target = spheres[k, 0:3] + (rng.random((len(k), 3), dtype=np.float32) * 2 - 1) * spheres[k, 3:4] * 1.5
    rays[k, 3:6] = (target - rays[k, 0:3]) * (rng.random((len(k), 1), dtype=np.float32) * 3 + 0.01)
    # exactly tangent in exact arithmetic: origin (c.x - 5, c.y + r, c.z), d = +x
    k = np.arange(6000, 6500)
    rays[k, 0:3] = spheres[k, 0:3] + np.stack([np.full(len(k), -5, np.float32), spheres[k, 3], np.zeros(len(k), np.float32)], 1)
    rays[k, 3:6] = np.array([1, 0, 0], np.float32)
    # origin inside
    k = np.arange(6500, 7200)
    rays[k, 0:3] = spheres[k, 0:3] + (rng.random((len(k), 3), dtype=np.float32) - 0.5) * spheres[k, 3:4]
    # sphere behind the ray
    k = np.arange(7200, 7700)
    rays[k, 3:6] = -(spheres[k, 0:3] - rays[k, 0:3])
    for k in range(7700, 8000):
        rays[k, rng.integers(0, 6)] = SPECIALS[rng.integers(0, len(SPECIALS))]
    got = p.api.probe_sphere(rays, spheres)
    exp = np.zeros(n, np.float32)
    O.lib().orc_sphere_batch(n, rays, spheres, exp)
    assert bits_equal(got, exp), mismatch_report(got, exp)
    assert 0.2 < (got < 3e38).mean() < 0.9


def _node_tree_scene(p, n=64, seed=5):
    """A bvh_node tree (bvh_node.cuh) over random spheres, built pairwise bottom-up through the API."""
    rng = np.random.default_rng(seed)
    s = p.Scene()
    refs = []
    for i in range(n):
        m = [s.Lambertian, lambda a: s.Metal(a, 0.3), lambda a: s.Dielectric((1, 1, 1), 1.5)][i % 3](rng.random(3, dtype=np.float32))
        c = (rng.random(3, dtype=np.float32) * 2 - 1) * 6
        refs.append(s.prim_ref(s.MakeSphere(c, float(rng.random() * 0.8 + 0.2), m)))
    while len(refs) > 1:
        nxt = [s.bvh_node(refs[i], refs[i + 1]) for i in range(0, len(refs) - 1, 2)]
        if len(refs) % 2:
            nxt.append(refs[-1])
        refs = nxt
    s.set_world_node_tree(refs[0])
    return s


@pytest.mark.parametrize("which", ["book1_final", "book2_moving", "three_spheres", "node_tree", "sah", "bottom_up"])
def test_world_closest_intersection_bit_exact(p, which):
    """G5: BVH::ClosestIntersection / HittableList / bvh_node over whole worlds: hit, t, primitive, normal."""
    if which == "node_tree":
        s = _node_tree_scene(p)
    elif which in ("sah", "bottom_up"):
        s = _node_tree_scene(p, n=80, seed=9)
        (s.BuildBVH_SAH if which == "sah" else s.BuildBVH_BottomUp)()
    else:
        s = config_scene(p, which)
    w = s.getWorldPtr()
    rng = np.random.default_rng(3)
    n = 16384
    rays = random_rays(rng, n, spread=12.0 if which in ("book1_final", "book2_moving") else 5.0)
    # half the rays start above the scene looking roughly down / towards the centre
    rays[: n // 2, 4] = -np.abs(rays[: n // 2, 4])
    for k in range(0, 256):  # axis-parallel and special components
        rays[k, 3 + rng.integers(0, 3)] = 0.0
    hit, t, prim, nrm = p.api.probe_trace(w, rays)
    ow = as_oracle_world(w)
    ehit = np.zeros(n, np.int32); et = np.zeros(n, np.float32); eprim = np.zeros(n, np.int32); en = np.zeros((n, 3), np.float32)
    assert O.lib().orc_trace_batch(C.byref(ow), n, rays, ehit, et, eprim, en) == 0
    assert np.array_equal(hit, ehit) and np.array_equal(prim, eprim), f"{(prim != eprim).sum()} primitive indices differ"
    assert bits_equal(t, et), mismatch_report(t, et)
    assert bits_equal(nrm, en), mismatch_report(nrm, en)
    assert 0.1 < hit.mean() < 0.99


def test_scatter_bit_exact(p):
    """G4: Scatter x {Lambertian, Metal fuzz 0 / 0.3 / 1, Dielectric front / back / TIR, checker}."""
    rng = np.random.default_rng(4)
    n = 8192
    mats = np.zeros(n, dtype=O.MAT_DT)
    mats["albedo"] = rng.random((n, 3), dtype=np.float32)
    mats["albedo2"] = rng.random((n, 3), dtype=np.float32)
    mats["type"] = np.arange(n) % 4
    metal = mats["type"] == 1
    mats["param"][metal] = rng.choice(np.array([0.0, 0.3, 1.0], np.float32), metal.sum())
    diel = mats["type"] == 2
    mats["param"][diel] = rng.choice(np.array([1.5, 1.0 / 1.5, 1.333, 2.4], np.float32), diel.sum())
    mats["param"][mats["type"] == 3] = np.float32(1.0) / np.float32(0.32)
    normals = rng.standard_normal((n, 3)).astype(np.float32)
    normals /= np.linalg.norm(normals, axis=1, keepdims=True).astype(np.float32)
    rays = random_rays(rng, n)
    rays[:, 3:6] *= rng.random((n, 1), dtype=np.float32) * 4 + 0.1       # un-normalised directions
    flip = rng.random(n) < 0.5                                            # front and back faces
    same = np.sum(rays[:, 3:6] * normals, axis=1) > 0
    normals[same != flip] *= -1
    # grazing incidence inside glass -> total internal reflection branch
    k = np.where(diel)[0][:400]
    tang = np.cross(normals[k], rng.standard_normal((len(k), 3)).astype(np.float32)).astype(np.float32)
    rays[k, 3:6] = tang + normals[k] * 0.05
    dist = (rng.random(n, dtype=np.float32) * 20).astype(np.float32)
    keys = rng.integers(0, 2**31, size=(n, 2), dtype=np.uint32)
    sc, orays, att, draws = p.api.probe_scatter(1984, mats, rays, dist, normals, keys)
    esc = np.zeros(n, np.int32); eor = np.zeros((n, 7), np.float32); eatt = np.zeros((n, 3), np.float32); edr = np.zeros(n, np.uint32)
    O.lib().orc_scatter_batch(1984, n, np.ascontiguousarray(mats).ctypes.data, rays, dist, normals, keys, esc, eor, eatt, edr)
    assert np.array_equal(sc, esc) and np.array_equal(draws, edr)
    assert bits_equal(orays, eor), mismatch_report(orays, eor)
    assert bits_equal(att, eatt), mismatch_report(att, eatt)
    assert 0 < sc[metal].mean() < 1  # both absorbed and reflected metal cases occur
    assert set(np.unique(draws[diel])) == {0, 1}  # uniform drawn only when refraction is possible (cu_materials.cuh:133)


@pytest.mark.parametrize("which", ["three_spheres", "book1_final", "book2_moving"])
def test_camera_sample_ray_bit_exact(p, which):
    """G3: sample_ray of the three cameras."""
    cam = config_cameras(p, which, 1200, 800)
    rng = np.random.default_rng(5)
    n = 4096
    st = (rng.random((n, 2), dtype=np.float32) * 2 - 1).astype(np.float32)
    keys = rng.integers(0, 2**31, size=(n, 2), dtype=np.uint32)
    rays, draws = p.api.probe_camera(1984, cam, st, keys)
    oc = as_oracle_camera(cam)
    er = np.zeros((n, 7), np.float32); ed = np.zeros(n, np.uint32)
    O.lib().orc_camera_batch(1984, C.byref(oc), n, st, keys, er, ed)
    assert np.array_equal(draws, ed)
    assert bits_equal(rays, er), mismatch_report(rays, er)


@pytest.mark.parametrize("which,W,H", [("book1_final", 1200, 800), ("book2_moving", 800, 800), ("three_spheres", 400, 225)])
def test_per_sample_radiance_bit_exact(p, which, W, H):
    """G6: one full sample (jitter, camera ray, <= 50 bounces) for 4096 (pixel, sample) keys."""
    s = config_scene(p, which)
    cam = config_cameras(p, which, W, H)
    w = s.getWorldPtr()
    rng = np.random.default_rng(6)
    n = 4096
    keys = np.stack([rng.integers(0, W * H, n), rng.integers(0, 1000, n)], axis=1).astype(np.uint32)
    cfg = p.capi.RenderConfig(W, H, 1, 50, 1984, 0, 0, 1, 0)
    got = p.api.probe_radiance(cfg, cam, w, keys)
    ow, oc = as_oracle_world(w), as_oracle_camera(cam)
    exp = np.zeros((n, 3), np.float32)
    assert O.lib().orc_radiance_batch(C.byref(ow), C.byref(oc), W, H, 50, 1984, n, keys, exp) == 0
    assert bits_equal(got, exp), mismatch_report(got, exp)
    assert got.max() > 0.5 and (got == 0).all(axis=1).mean() < 0.9


def test_reference_gtest_twin_sphere_index(p):
    """google_testing/test.cpp SphereTest.DeviceSphereIndexTest: nearest-sphere index per pixel, host
    ground truth vs device kernel, exact equality at 1280x720 over the 488-sphere layout."""
    s = p.Scene.book1_final(1984)
    _, prims, _ = s.arrays()
    spheres = np.concatenate([prims["c0"], prims["radius"][:, None]], axis=1).astype(np.float32)
    W, H = 1280, 720
    cam = p.PinholeCamera((0, 1, -4), (0, 1, 0), (0, 1, 0), 90.0, W / H)  # test.cpp:21-26
    got = p.api.probe_sphere_index(cam, W, H, spheres)
    oc = as_oracle_camera(cam)
    exp = np.zeros(W * H, np.int32)
    O.lib().orc_sphere_index(C.byref(oc), W, H, len(spheres), spheres, exp)
    assert np.array_equal(got.ravel(), exp)
    assert len(np.unique(got)) > 50 and (got == -1).mean() > 0.2


# ------------------------------------------------------------------------------------------------
# framebuffer parity
# ------------------------------------------------------------------------------------------------
def _render_gpu(p, which, W, H, spp, depth=50, variant=0, seed=1984):
    s = config_scene(p, which, seed)
    cam = config_cameras(p, which, W, H)
    r = p.Renderer.MakeRenderer(W, H, spp, depth, cam, s.getWorldPtr(), seed=seed, variant=variant)
    r.Render()
    img = r.DownloadRenderbuffer()
    ms = r.last_kernel_ms()
    r.close()
    return img, ms


def _render_cpu(which, W, H, spp, depth=50, seed=1984):
    o = oracle_scene(which, seed)
    if which == "three_spheres":
        cam = O.camera_pinhole((0, 0, 0), (0, 0, -1), (0, 1, 0), 90.0, W / H)
    elif which == "book1_final":
        cam = O.camera_defocus((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, W / H, 0.1, 10.0)
    else:
        cam = O.camera_motion((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, W / H, 0.0, 1.0)
    img, cnt = O.render(o.world, cam, W, H, spp, depth, seed)
    return img, cnt


VARIANTS = [1, 2, 3, 4, 5]   # 1 = baseline wave-per-pixel, 2 = streaming LDS kernel, 3 = 2 + fast exact division, 4 = 3 + filtered predicates, 5 = 3 with rays exchanged between the waves of a workgroup
BIT_EXACT_VARIANTS = {2, 3, 4, 5}  # the streaming path sums samples in the reference's order: image == oracle image


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("which,W,H,spp", [
    ("three_spheres", 400, 225, 1),    # BASELINE.json configs[0] at full size
    ("three_spheres", 100, 57, 67),    # ragged size (not a multiple of the 8x8 tile), odd spp
    ("book1_final", 300, 200, 16),     # configs[1] scene, reduced
    ("book1_final", 61, 43, 130),      # ragged, spp > 2 wavefronts
    ("book2_moving", 200, 200, 24),    # configs[2] scene, reduced
])
def test_framebuffer_matches_oracle(p, which, W, H, spp, variant):
    if which == "three_spheres" and variant >= 3:
        pytest.skip("the fast-division variants take RT_WORLD_BVH worlds; a HittableList renders on variant 2")
    img, _ = _render_gpu(p, which, W, H, spp, variant=variant)
    ref, _ = _render_cpu(which, W, H, spp)
    assert img.shape == ref.shape == (H, W, 4)
    assert np.all(img[..., 3] == 1.0)
    d = np.abs(img - ref)
    assert np.nanmax(d) < TOL_SPEC, f"max |delta| {np.nanmax(d)}"
    assert np.nanmax(d) <= TOL_MEASURED, f"max |delta| {np.nanmax(d)} exceeds the summation-order bound"
    assert np.array_equal(np.isnan(img), np.isnan(ref))
    if variant in BIT_EXACT_VARIANTS:
        assert bits_equal(img, ref), mismatch_report(img, ref)


@pytest.mark.parametrize("variant", VARIANTS)
def test_spp1_framebuffer_is_bit_exact(p, variant):
    """With one sample per pixel there is no summation: the image must equal the oracle's bit for bit."""
    img, _ = _render_gpu(p, "book1_final", 320, 200, 1, variant=variant)
    ref, _ = _render_cpu("book1_final", 320, 200, 1)
    assert bits_equal(img, ref), mismatch_report(img, ref)


def test_max_depth_is_honoured(p):
    """The committed reference app renders with max_depth 4 (FirstApp.cpp:39)."""
    img, _ = _render_gpu(p, "book2_moving", 160, 90, 8, depth=4)
    ref, _ = _render_cpu("book2_moving", 160, 90, 8, depth=4)
    assert np.nanmax(np.abs(img - ref)) <= TOL_MEASURED
    img0, _ = _render_gpu(p, "book2_moving", 32, 16, 2, depth=0)
    assert np.all(img0[..., :3] == 0.0)  # max bounces exceeded -> black (Renderer.cu:178-180)


def _sparse_gids(img, rng, n):
    """n random pixels plus EVERY non-finite pixel of the GPU image.  A NaN pixel is legitimate: the
    reference's dielectric returns a zero direction when `refract`'s k < 0 disagrees by one rounding with
    the `ior_ratio * sin_theta > 1` test (cu_materials.cuh:133-137, func_geometric.inl:117-120), and the
    sky term then normalises a zero vector (Renderer.cu:150).  At ~1e-9 per sample it shows up only in
    full-size frames; the oracle must produce NaN at exactly the same pixels."""
    H, W = img.shape[:2]
    bad = np.argwhere(~np.isfinite(img[..., :3]).all(axis=2))
    assert len(bad) <= 16, f"{len(bad)} non-finite pixels"
    extra = (bad[:, 0] * W + bad[:, 1]).astype(np.uint32)
    fin = img[np.isfinite(img)]
    assert fin.min() >= 0.0 and fin.max() <= 1.0
    return np.concatenate([rng.integers(0, W * H, n).astype(np.uint32), extra])


def _check_sparse(got, exp):
    assert np.array_equal(np.isnan(got), np.isnan(exp)), "NaN pixels differ between GPU and oracle"
    d = float(np.nanmax(np.abs(got - exp)))
    assert d < TOL_SPEC and d <= TOL_MEASURED, d
    return d


@pytest.mark.parametrize("variant", VARIANTS)
def test_full_size_config2_properties(p, variant):
    """BASELINE.json configs[1] at FULL size (1200x800, 500 spp = 4.8e8 samples) through properties:
    determinism, range, alpha, and exact agreement with the oracle on a sparse sample of pixels
    (the oracle renders just those pixels, all 500 samples each)."""
    W, H, spp = 1200, 800, 500
    img, ms = _render_gpu(p, "book1_final", W, H, spp, variant=variant)
    img2, _ = _render_gpu(p, "book1_final", W, H, spp, variant=variant)
    assert img.tobytes() == img2.tobytes(), "render is not deterministic"
    assert np.all(img[..., 3] == 1.0)
    rng = np.random.default_rng(11)
    gids = _sparse_gids(img, rng, 192)
    o = oracle_scene("book1_final")
    oc = O.camera_defocus((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, W / H, 0.1, 10.0)
    exp = O.render_pixels(o.world, oc, W, H, spp, 50, gids)
    got = img.reshape(-1, 4)[gids]
    d = _check_sparse(got, exp)
    if variant in BIT_EXACT_VARIANTS:
        assert bits_equal(got, exp), mismatch_report(got, exp)
    print(f"config2 full size variant {variant}: {W*H*spp/ms/1e3:.1f} Msamples/s ({ms:.1f} ms), max|delta| on 192 px = {d:.2e}")


def test_full_size_config3_sparse_parity(p):
    """configs[2]: Book-2 moving spheres 800x800x1000 — sparse exact check at full size."""
    W, H, spp = 800, 800, 1000
    img, ms = _render_gpu(p, "book2_moving", W, H, spp)
    rng = np.random.default_rng(12)
    gids = _sparse_gids(img, rng, 96)
    o = oracle_scene("book2_moving")
    oc = O.camera_motion((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, W / H, 0.0, 1.0)
    exp = O.render_pixels(o.world, oc, W, H, spp, 50, gids)
    got = img.reshape(-1, 4)[gids]
    _check_sparse(got, exp)
    assert bits_equal(got, exp), mismatch_report(got, exp)  # default variant = streaming path
    assert np.all(img[..., 3] == 1.0)
    print(f"config3 full size: {W*H*spp/ms/1e3:.1f} Msamples/s ({ms:.1f} ms)")


@pytest.mark.parametrize("world_size", [2, 3, 8])
def test_tile_sharding_is_gpu_count_invariant(p, world_size):
    """Tile-shard the frame over `world_size` ranks (all run on this one GPU, one after another),
    concatenate the shards rank-major as the RCCL gather would, assemble, and require the SAME BITS as
    the single-GPU image: counter-based RNG keys depend only on (pixel, sample)."""
    import torch
    W, H, spp = 203, 117, 20  # ragged: 26 x 15 tiles, not divisible by 8
    s = config_scene(p, "book1_final")
    cam = config_cameras(p, "book1_final", W, H)
    w = s.getWorldPtr()
    single = p.Renderer.MakeRenderer(W, H, spp, 50, cam, w)
    single.Render()
    ref = single.DownloadRenderbuffer()
    shards = []
    for rank in range(world_size):
        r = p.Renderer.MakeRenderer(W, H, spp, 50, cam, w, rank=rank, world_size=world_size)
        n = r.shard_floats()
        buf = torch.zeros(n, dtype=torch.float32, device="cuda:0")
        r.render_async(torch.cuda.current_stream().cuda_stream, buf.data_ptr())
        torch.cuda.synchronize()
        shards.append(buf)
        last = r
    gathered = torch.cat(shards)
    image = torch.empty(H * W * 4, dtype=torch.float32, device="cuda:0")
    last.assemble(gathered.data_ptr(), image.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = image.cpu().numpy().reshape(H, W, 4)
    assert got.tobytes() == ref.tobytes()


def test_download_after_render_async_on_a_side_stream_is_ordered(p):
    """rt_renderer_render_async(r, caller_stream, NULL) renders into the renderer's own framebuffer; DownloadRenderbuffer must wait for
    THAT stream's work (the renderer's own stream is non-blocking, so syncing it alone would copy a stale or partial frame)."""
    import torch
    W, H, spp = 400, 300, 40
    s = config_scene(p, "book1_final")
    cam = config_cameras(p, "book1_final", W, H)
    r = p.Renderer.MakeRenderer(W, H, spp, 50, cam, s.getWorldPtr())
    r.Render()
    ref = r.DownloadRenderbuffer()
    r.close()
    r = p.Renderer.MakeRenderer(W, H, spp, 50, cam, s.getWorldPtr())
    side = torch.cuda.Stream()
    r.render_async(side.cuda_stream, None)
    got = r.DownloadRenderbuffer()          # no explicit synchronisation by the caller
    r.close()
    assert got.tobytes() == ref.tobytes()
    # rt_renderer_kernel_times: per-kernel HIP-event durations of the last renders (what bench.py prices the roofline with)
    r = p.Renderer.MakeRenderer(W, H, spp, 50, cam, s.getWorldPtr())
    with pytest.raises(p.capi.RtError):
        r.kernel_times(0)
    r.Render(); r.Render()
    t_primary, t_stream, t_resolve = r.kernel_times(0)
    assert 0 < t_primary < t_stream and 0 < t_resolve < t_stream
    assert abs(sum(r.kernel_times(1)) - sum(r.kernel_times(0))) < 0.5 * sum(r.kernel_times(0))
    with pytest.raises(p.capi.RtError):
        r.kernel_times(2)
    r.close()


def test_bad_world_is_refused_not_faulted(p):
    """Indices are validated on the host before any kernel follows them."""
    s = p.Scene.book1_final(1)
    w = s.getWorldPtr()
    nodes, _, _ = s.arrays()
    nodes = nodes.copy()
    nodes[10]["left"] = 5000
    w.nodes = nodes.ctypes.data
    cam = config_cameras(p, "book1_final", 64, 64)
    with pytest.raises(p.capi.RtError, match="out of range"):
        p.Renderer.MakeRenderer(64, 64, 1, 4, cam, w)
    with pytest.raises(p.capi.RtError, match="must be > 0"):
        p.Renderer.MakeRenderer(0, 64, 1, 4, cam, s.getWorldPtr())


def test_multi_pass_rendering_is_bit_identical(p, monkeypatch):
    """When spp does not fit the per-pass sample buffer the frame is rendered in several passes; the
    resolve kernel carries the running sum in sample order, so the image must not change by one bit."""
    W, H, spp = 96, 64, 37
    s = config_scene(p, "book2_moving")
    cam = config_cameras(p, "book2_moving", W, H)
    r = p.Renderer.MakeRenderer(W, H, spp, 50, cam, s.getWorldPtr(), variant=2)
    r.Render()
    one = r.DownloadRenderbuffer()
    r.close()
    monkeypatch.setenv("RT06_PASS_SPP", "5")  # 5 samples per pixel per pass -> 8 passes
    r = p.Renderer.MakeRenderer(W, H, spp, 50, cam, s.getWorldPtr(), variant=2)
    r.Render()
    many = r.DownloadRenderbuffer()
    r.close()
    assert one.tobytes() == many.tobytes()
    ref, _ = _render_cpu("book2_moving", W, H, spp)
    assert bits_equal(one, ref), mismatch_report(one, ref)


# ------------------------------------------------------------------------------------------------
# the 5-instruction correctly rounded division of the streaming kernel (csrc/rt_fastdiv.hpp)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("four", [False, True])
def test_fastdiv_matches_ieee_division(p, four):
    """q2 == n/d bit for bit: 64 divisor significands (incl. the all-ones and power-of-two ones) x ALL 2^23
    numerator significands, at the centre and at the corners of the regular class's exponent range — for the
    5-instruction form (root box, filtered variant) and the 4-instruction two-word-reciprocal form (hot loop).
    tools/verify_fastdiv.py [--four] sweeps every divisor significand (2^46 pairs); results in profiles/."""
    rng = np.random.default_rng(21)
    special = [0, 1, 2, 3, (1 << 23) - 1, (1 << 23) - 2, 1 << 22, (1 << 22) - 1, (1 << 22) + 1, 0x2AAAAA, 0x555555]
    firsts = special + rng.integers(0, 1 << 23, 53).tolist()
    total = 0
    for k, first in enumerate(firsts):
        ne, de = [(0, 0), (-64, 39), (40, -40), (-64, -40), (40, 39)][k % 5]
        bad, ex = p.api.selftest_fastdiv(int(first), 1, ne, de, four=four)
        assert bad == 0, f"divisor significand {first:#x}: {bad} mismatches, e.g. n={ex[0]:#x} d={ex[1]:#x}"
        total += 1 << 23
    assert total == 64 << 23


def test_fast_reciprocal_is_the_ieee_reciprocal_for_every_value_of_the_class(p):
    """RN(1/d) of a regular ray is hardware rcp + one Newton step; exhaustive over all fp32 with 2^-40 <= |x| < 2^40."""
    n, bad, ex = p.api.selftest_fastrcp()
    assert n == 2 * 80 * (1 << 23)
    assert bad == 0, f"{bad} mismatches, e.g. x={ex:#x}"


def test_box_test_with_fast_division_makes_identical_decisions(p):
    """aabb_intersects_regular vs aabb::intersects on regular rays: same hit flag and same dist."""
    rng = np.random.default_rng(22)
    n = 1 << 18
    lo = (rng.random((n, 3), dtype=np.float32) * 2 - 1) * 12
    ext = rng.random((n, 3), dtype=np.float32) * 3
    boxes = np.concatenate([lo, lo + ext], axis=1).astype(np.float32)
    boxes[: n // 8, 1] = 0.0  # exact-zero coordinates (ground-level boxes), like the book scenes
    rays = random_rays(rng, n, with_time=False)
    rays[n // 2:, 0:3] = lo[n // 2:] + ext[n // 2:] * rng.random((n // 2, 3), dtype=np.float32) * 1.5  # near / inside the box
    rays[: n // 16, 3] *= np.float32(1e-6)   # near axis-parallel but still regular
    rays[n // 16: n // 8, 4] = 0.0           # irregular: exact zero direction component
    maxd = np.where(rng.random(n) < 0.5, np.float32(3.402823466e38), rng.random(n, dtype=np.float32) * 30).astype(np.float32)
    reg, hit, dist = p.api.probe_aabb_regular(boxes, rays, maxd)
    ehit = np.zeros(n, np.int32); edist = np.zeros(n, np.float32)
    O.lib().orc_aabb_batch(n, boxes, rays, maxd, ehit, edist)
    m = reg == 1
    assert 0.8 < m.mean() < 0.99 and not reg[n // 16: n // 8].any()
    assert np.array_equal(hit[m], ehit[m])
    # dist may differ in the sign of a zero only (IEEE min/max vs GLM's (y<x)?y:x)
    assert np.array_equal(dist[m] + np.float32(0.0), edist[m] + np.float32(0.0))
    assert bits_equal(np.abs(dist[m]), np.abs(edist[m]))


def test_filtered_box_pair_predicates_are_sound(p):
    """Whenever box_pair_filtered does not say `uncertain`, its three decisions equal the exact ones —
    on random pairs and on adversarial near-ties (shared faces, rays through edges and corners, rec_t on the
    entry distance)."""
    rng = np.random.default_rng(23)
    n = 1 << 20
    lo = ((rng.random((n, 3), dtype=np.float32) * 2 - 1) * 12).astype(np.float32)
    ext = (rng.random((n, 3), dtype=np.float32) * 3 + 0.01).astype(np.float32)
    left = np.concatenate([lo, lo + ext], axis=1)
    # right box: a sibling that shares faces with the left one on some axes
    shift = np.where(rng.random((n, 3)) < 0.5, 0.0, rng.random((n, 3)) * 2 - 1).astype(np.float32)
    rlo = (lo + shift * ext).astype(np.float32)
    rext = np.where(rng.random((n, 3)) < 0.5, ext, (rng.random((n, 3), dtype=np.float32) * 3 + 0.01)).astype(np.float32)
    right = np.concatenate([rlo, rlo + rext], axis=1)
    boxes = np.ascontiguousarray(np.concatenate([left, right], axis=1), dtype=np.float32)
    rays = random_rays(rng, n, with_time=False)
    # aim a quarter of the rays exactly at a corner / edge point of the left box
    k = np.arange(0, n // 4)
    corner = np.where(rng.random((len(k), 3)) < 0.5, left[k, 0:3], left[k, 3:6]).astype(np.float32)
    rays[k, 3:6] = (corner - rays[k, 0:3]) * (rng.random((len(k), 1), dtype=np.float32) + 0.5)
    maxd = np.where(rng.random(n) < 0.5, np.float32(3.402823466e38), rng.random(n, dtype=np.float32) * 30).astype(np.float32)
    # rec_t exactly on the left box's entry distance for some cases
    hit, dist = p.api.probe_aabb(np.ascontiguousarray(boxes[:, 0:6]), rays, np.full(n, 3.402823466e38, np.float32))
    kk = np.where(hit[: n // 8] == 1)[0]
    maxd[kk] = dist[kk]
    out = p.api.probe_boxpair_filtered(boxes, rays, maxd)
    reg = out[:, 0] == 1
    sure = reg & (out[:, 1] == 0)
    assert reg.mean() > 0.95
    assert np.array_equal(out[sure, 2], out[sure, 5]), "hit_left differs"
    assert np.array_equal(out[sure, 3], out[sure, 6]), "hit_right differs"
    assert np.array_equal(out[sure, 4], out[sure, 7]), "near/far order differs"
    unc_rate = out[reg, 1].mean()
    print(f"filtered predicates: uncertain on {unc_rate:.2e} of {reg.sum()} adversarial visits")
    assert unc_rate < 0.4  # adversarial set (a quarter of the rays aim exactly at a box corner); ~1e-6 in real traversals
    assert out[sure, 2].mean() > 0.02 and out[sure, 4].mean() > 0.005


def test_certified_far_planes_take_the_exact_decisions(p):
    """The default hot loop's box pair (rt_fastdiv.hpp, CERTIFIED FAR PLANES): near parameters exact, far parameters products with the rounded
    reciprocal, `tmin <= tmax` certified or redone exactly.  Its three decisions must equal aabb::intersects' on EVERY regular case — random pairs,
    sibling boxes that share faces, rays aimed exactly at corners and edge points of a box (tmin == tmax up to rounding: the cases the certification
    exists for), flat boxes (tmin == tmax on an axis), rays starting on a face, rec_t on the entry distance."""
    rng = np.random.default_rng(29)
    n = 1 << 21
    lo = ((rng.random((n, 3), dtype=np.float32) * 2 - 1) * 12).astype(np.float32)
    ext = (rng.random((n, 3), dtype=np.float32) * 3 + 0.01).astype(np.float32)
    flat = rng.random((n, 3)) < 0.03
    ext[flat] = 0.0                                   # zero-thickness boxes on some axes: near == far plane there
    left = np.concatenate([lo, lo + ext], axis=1)
    shift = np.where(rng.random((n, 3)) < 0.5, 0.0, rng.random((n, 3)) * 2 - 1).astype(np.float32)
    rlo = (lo + shift * ext).astype(np.float32)
    rext = np.where(rng.random((n, 3)) < 0.5, ext, (rng.random((n, 3), dtype=np.float32) * 3 + 0.01)).astype(np.float32)
    right = np.concatenate([rlo, rlo + rext], axis=1)
    boxes = np.ascontiguousarray(np.concatenate([left, right], axis=1), dtype=np.float32)
    rays = random_rays(rng, n, with_time=False)
    q = n // 4
    # a quarter: aimed exactly at a corner of the left box; another quarter: at a point of an EDGE (two coordinates on planes, one inside)
    corner = np.where(rng.random((q, 3)) < 0.5, left[:q, 0:3], left[:q, 3:6]).astype(np.float32)
    rays[:q, 3:6] = (corner - rays[:q, 0:3]) * (rng.random((q, 1), dtype=np.float32) + 0.5)
    edge = np.where(rng.random((q, 3)) < 0.5, right[q:2 * q, 0:3], right[q:2 * q, 3:6]).astype(np.float32)
    free = rng.integers(0, 3, q)
    t = rng.random(q, dtype=np.float32)
    edge[np.arange(q), free] = (right[q:2 * q, 0:3][np.arange(q), free] * (1 - t) + right[q:2 * q, 3:6][np.arange(q), free] * t).astype(np.float32)
    rays[q:2 * q, 3:6] = (edge - rays[q:2 * q, 0:3]) * (rng.random((q, 1), dtype=np.float32) * 3 + 0.25)
    # an eighth: the origin ON a face plane of the left box (a zero plane offset: t == 0 exactly)
    k = np.arange(2 * q, 2 * q + n // 8)
    ax = rng.integers(0, 3, len(k))
    rays[k, ax] = np.where(rng.random(len(k)) < 0.5, left[k, ax], left[k, 3 + ax])
    maxd = np.where(rng.random(n) < 0.5, np.float32(3.402823466e38), rng.random(n, dtype=np.float32) * 30).astype(np.float32)
    hit, dist = p.api.probe_aabb(np.ascontiguousarray(boxes[:, 0:6]), rays, np.full(n, 3.402823466e38, np.float32))
    kk = np.where(hit[: n // 8] == 1)[0]
    maxd[kk] = dist[kk]                                # rec_t exactly on the left box's entry distance
    out = p.api.probe_boxpair_certified(boxes, rays, maxd)
    reg = out[:, 0] == 1
    assert reg.mean() > 0.9
    assert np.array_equal(out[reg, 2], out[reg, 5]), f"hit_left differs in {(out[reg, 2] != out[reg, 5]).sum()} cases"
    assert np.array_equal(out[reg, 3], out[reg, 6]), f"hit_right differs in {(out[reg, 3] != out[reg, 6]).sum()} cases"
    assert np.array_equal(out[reg, 4], out[reg, 7]), f"near/far order differs in {(out[reg, 4] != out[reg, 7]).sum()} cases"
    unc = out[reg, 1].mean()
    tail = np.zeros(n, bool)
    tail[3 * q + n // 8:] = True                       # the random tail of the set, boxes of non-zero thickness: what a traversal sees
    tail &= reg & ~flat.any(axis=1) & (rext > 0).all(axis=1)
    plain = out[tail, 1].mean()
    print(f"certified far planes: exact redo on {unc:.2e} of {reg.sum()} adversarial visits, {plain:.2e} of the random ones")
    assert unc > 1e-4        # the adversarial set does reach the fallback (corner / edge rays), so the test covers it
    assert plain < 1e-4      # random visits almost never do
    assert out[reg, 2].mean() > 0.02 and out[reg, 4].mean() > 0.005


# ------------------------------------------------------------------------------------------------
# world shapes and materials beyond the prefab scenes
# ------------------------------------------------------------------------------------------------
def _render_both(p, scene, cam, W, H, spp, depth=50, variant=0):
    w = scene.getWorldPtr()
    r = p.Renderer.MakeRenderer(W, H, spp, depth, cam, w, variant=variant)
    r.Render()
    img = r.DownloadRenderbuffer()
    r.close()
    ref, _ = O.render(as_oracle_world(w), as_oracle_camera(cam), W, H, spp, depth)
    return img, ref


@pytest.mark.parametrize("W,H,spp,depth", [(1, 1, 1, 1), (1, 1, 7, 50), (1, 9, 3, 50), (9, 1, 3, 50), (8, 8, 1, 2), (7, 7, 65, 50), (17, 1, 129, 4)])
def test_smallest_frames_render_bit_exact(p, W, H, spp, depth):
    """the smallest inputs the boundary accepts — one pixel, one row, one column, exactly one 8x8 tile, less than a tile — on the streaming kernel
    (a pass of 1 .. 1105 sample indices, mostly padding) and on the baseline kernel: the oracle's frame"""
    scene = p.Scene.book1_final(1984)
    cam = p.DefocusBlurCamera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, W / H, 0.1, 10.0)
    for variant in (0, 1):
        img, ref = _render_both(p, scene, cam, W, H, spp, depth, variant=variant)
        assert img.shape == (H, W, 4)
        if variant == 0:
            assert bits_equal(img, ref), mismatch_report(img, ref)
        else:   # (the baseline kernel sums a pixel's samples in another order)
            assert np.nanmax(np.abs(img - ref)) <= 1e-5


@pytest.mark.parametrize("n_spheres", [1, 2, 3])
def test_tiny_bvh_worlds_render_bit_exact(p, n_spheres):
    """A BVH whose root is a leaf (one sphere) or has leaf children only — the streaming kernel's degenerate trees."""
    s = p.Scene()
    for i in range(n_spheres):
        m = [s.Lambertian((0.8, 0.3, 0.3)), s.Metal((0.8, 0.8, 0.8), 0.2), s.Dielectric((1, 1, 1), 1.5)][i]
        s.MakeSphere((1.2 * i - 1.0, 0.0, -2.0 - 0.3 * i), 0.5, m)
    s.BuildBVH_TopDown()
    cam = p.PinholeCamera((0, 0, 0), (0, 0, -1), (0, 1, 0), 70.0, 96 / 64)
    img, ref = _render_both(p, s, cam, 96, 64, 9)
    assert bits_equal(img, ref), mismatch_report(img, ref)
    assert (img[..., :3].std() > 0.01)


def test_checker_texture_and_moving_spheres_render_bit_exact(p):
    """LambertianTexture (cu_materials.cuh:16-41, checker_texture cu_Textures.cuh:21-40) on static and moving spheres,
    through the streaming kernel (second colour read from global memory) and the baseline kernel."""
    rng = np.random.default_rng(31)
    s = p.Scene()
    ground = s.LambertianTexture((0.2, 0.3, 0.1), (0.9, 0.9, 0.9), 0.32)
    s.MakeSphere((0, -1000, 0), 1000.0, ground)
    for i in range(24):
        c = ((rng.random() * 2 - 1) * 4, 0.3, (rng.random() * 2 - 1) * 4)
        m = s.LambertianTexture(rng.random(3), rng.random(3), 0.1) if i % 2 else s.Metal(rng.random(3), 0.1)
        if i % 3 == 0:
            s.MakeMovingSphere(c, (c[0], c[1] + 0.3, c[2]), 0.3, m)
        else:
            s.MakeSphere(c, 0.3, m)
    s.BuildBVH_TopDown()
    cam = p.MotionBlurCamera((6, 2, 5), (0, 0, 0), (0, 1, 0), 40.0, 1.5, 0.0, 1.0)
    for variant in (1, 2, 3):
        img, ref = _render_both(p, s, cam, 120, 80, 11, variant=variant)
        if variant == 1:
            assert np.nanmax(np.abs(img - ref)) <= TOL_MEASURED
        else:
            assert bits_equal(img, ref), mismatch_report(img, ref)


def test_hittable_list_and_bvh_node_worlds_render(p):
    """HittableList (HittableList.cuh:21-34) and bvh_node (bvh_node.cuh:19-24) worlds: bit-exact on the streaming
    kernel (their own traversal modes in LDS), summation-order-close on the baseline kernel."""
    s = _node_tree_scene(p, n=40, seed=3)
    cam = p.PinholeCamera((0, 3, 14), (0, 0, 0), (0, 1, 0), 50.0, 1.5)
    img, ref = _render_both(p, s, cam, 96, 64, 5)
    assert bits_equal(img, ref), mismatch_report(img, ref)
    img_b, _ = _render_both(p, s, cam, 96, 64, 5, variant=1)
    assert np.nanmax(np.abs(img_b - ref)) <= TOL_MEASURED
    with pytest.raises(p.capi.RtError, match="RT_WORLD_BVH"):
        p.Renderer.MakeRenderer(96, 64, 5, 50, cam, s.getWorldPtr(), variant=3)
    s.MakeHittableList()
    img2, ref2 = _render_both(p, s, cam, 96, 64, 5)
    assert bits_equal(img2, ref2), mismatch_report(img2, ref2)
    img2_b, _ = _render_both(p, s, cam, 96, 64, 5, variant=1)
    assert np.nanmax(np.abs(img2_b - ref2)) <= TOL_MEASURED
    # the three world kinds over the same spheres give the same closest hits, hence the same image
    s.BuildBVH_SAH()
    img3, ref3 = _render_both(p, s, cam, 96, 64, 5)
    assert bits_equal(img3, ref3)
    assert img3.tobytes() == img2.tobytes() == img.tobytes()


def test_irregular_box_coordinates_fall_back_to_the_verbatim_kernel(p):
    """A box coordinate outside the fast-division class (here 1e-15) must not be rendered by variant 3."""
    s = p.Scene()
    m = s.Lambertian((0.5, 0.5, 0.5))
    s.MakeSphere((0.5, 0.0, -2.0), 0.5, m)
    s.MakeSphere((2.0, 0.0, -3.0), 0.5, m)
    s.MakeSphere((2e-15, 0.4, -1.0), 1e-15, m)  # box min.x = 1e-15: not 0 and below 2^-40
    s.BuildBVH_TopDown()
    nodes, _, _ = s.arrays()
    assert np.any((np.abs(nodes["min"]) > 0) & (np.abs(nodes["min"]) < 2.0 ** -40))
    cam = p.PinholeCamera((0, 0, 0), (0, 0, -1), (0, 1, 0), 70.0, 1.5)
    with pytest.raises(p.capi.RtError, match="box coordinate"):
        p.Renderer.MakeRenderer(48, 32, 2, 8, cam, s.getWorldPtr(), variant=3)
    img, ref = _render_both(p, s, cam, 48, 32, 2, depth=8)  # default resolves to the verbatim streaming kernel
    assert bits_equal(img, ref), mismatch_report(img, ref)


def test_rays_outside_the_fast_division_class_take_the_verbatim_loop(p):
    """A camera origin component of 1e-30 (non-zero, below 2^-60) puts EVERY primary ray outside the fast-division
    class of rt_fastdiv.hpp while the scattered rays are inside it: the default kernel must step the former through
    its verbatim loop (marked inner references) and the latter through the fast one, in the same wave."""
    s = config_scene(p, "book1_final")
    W, H, spp = 160, 100, 8
    cam = p.DefocusBlurCamera((1e-30, 2.0, 9.0), (0, 0.5, 0), (0, 1, 0), 35.0, W / H, 0.0, 9.0)
    img, ref = _render_both(p, s, cam, W, H, spp)
    assert bits_equal(img, ref), mismatch_report(img, ref)
    cam0 = p.DefocusBlurCamera((0.0, 2.0, 9.0), (0, 0.5, 0), (0, 1, 0), 35.0, W / H, 0.0, 9.0)  # zero IS in the class
    img0, ref0 = _render_both(p, s, cam0, W, H, spp)
    assert bits_equal(img0, ref0), mismatch_report(img0, ref0)
    # axis-parallel view direction: primary rays with exactly zero direction components (1/d = inf)
    cam1 = p.PinholeCamera((0, 1, 12), (0, 1, 0), (0, 1, 0), 40.0, W / H)
    img1, ref1 = _render_both(p, s, cam1, W, H, 1)
    assert bits_equal(img1, ref1), mismatch_report(img1, ref1)


@pytest.mark.parametrize("seed", list(range(12)))
def test_random_scenes_cameras_and_builders_render_bit_exact(p, seed):
    """Differential fuzz: random sphere soups (static / moving, all four materials, overlapping and nested spheres,
    a camera possibly INSIDE a glass sphere), random camera type, resolution, spp, depth and world builder —
    GPU framebuffer vs oracle, bit for bit."""
    rng = np.random.default_rng(1000 + seed)
    s = p.Scene()
    n = int(rng.integers(1, 60))
    for i in range(n):
        kind = int(rng.integers(0, 4))
        col = rng.random(3, dtype=np.float32)
        if kind == 0:
            m = s.Lambertian(col)
        elif kind == 1:
            m = s.Metal(col, float(rng.choice([0.0, 0.05, 0.5, 1.0])))
        elif kind == 2:
            m = s.Dielectric((1, 1, 1), float(rng.choice([1.5, 1.0 / 1.5, 1.33, 2.4])))
        else:
            m = s.LambertianTexture(col, rng.random(3, dtype=np.float32), float(rng.choice([0.1, 0.32, 1.0])))
        c = ((rng.random(3) * 2 - 1) * np.array([6, 2, 6])).astype(np.float32)
        r = float(rng.choice([0.05, 0.3, 0.8, 2.5]))
        if rng.random() < 0.3:
            s.MakeMovingSphere(c, c + (rng.random(3).astype(np.float32) - 0.5), r, m)
        else:
            s.MakeSphere(c, r, m)
    if rng.random() < 0.5:
        s.MakeSphere((0, -500.0, 0), 498.0, s.Lambertian((0.5, 0.5, 0.5)))
    builder = int(rng.integers(0, 4))
    [s.BuildBVH_TopDown, s.BuildBVH_SAH, s.BuildBVH_BottomUp, s.MakeHittableList][builder]()
    W, H = int(rng.integers(17, 97)), int(rng.integers(9, 65))
    spp, depth = int(rng.integers(1, 24)), int(rng.choice([1, 2, 5, 50]))
    eye = ((rng.random(3) * 2 - 1) * np.array([8, 3, 8])).astype(np.float32)
    cam_kind = int(rng.integers(0, 3))
    if cam_kind == 0:
        cam = p.PinholeCamera(eye, (0, 0, 0), (0, 1, 0), float(rng.uniform(20, 100)), W / H)
    elif cam_kind == 1:
        cam = p.DefocusBlurCamera(eye, (0, 0, 0), (0, 1, 0), float(rng.uniform(20, 100)), W / H, float(rng.uniform(0, 0.5)), float(rng.uniform(2, 12)))
    else:
        cam = p.MotionBlurCamera(eye, (0, 0, 0), (0, 1, 0), float(rng.uniform(20, 100)), W / H, 0.0, 1.0)
    img, ref = _render_both(p, s, cam, W, H, spp, depth=depth)
    assert np.array_equal(np.isnan(img), np.isnan(ref))
    assert bits_equal(img, ref), f"seed {seed} builder {builder} cam {cam_kind} {W}x{H}x{spp} depth {depth}: " + mismatch_report(img, ref)


# ------------------------------------------------------------------------------------------------
# worlds that do not fit the LDS: same streaming kernel, records from global memory, 32-bit references
# ------------------------------------------------------------------------------------------------
def _big_sphere_scene(p, n, seed, lights_and_quads=False):
    rng = np.random.default_rng(seed)
    s = p.Scene()
    mats = [s.Lambertian((0.7, 0.3, 0.3)), s.Metal((0.8, 0.8, 0.9), 0.1), s.Dielectric((1, 1, 1), 1.5),
            s.LambertianTexture((0.2, 0.3, 0.1), (0.9, 0.9, 0.9), 0.32), s.Lambertian((0.3, 0.3, 0.8))]
    if lights_and_quads:
        mats.append(s.DiffuseLight((4.0, 4.0, 3.0)))
    s.MakeSphere((0, -1000.0, 0), 1000.0, mats[3])
    for i in range(n):
        c = ((rng.random(3) * 2 - 1) * np.array([30, 0, 30]) + np.array([0, 0.2 + rng.random() * 3, 0])).astype(np.float32)
        r = float(0.05 + rng.random() * 0.25)
        m = mats[int(rng.integers(0, len(mats)))]
        if i % 7 == 0:
            s.MakeMovingSphere(c, c + np.float32([0, 0.3, 0]), r, m)
        else:
            s.MakeSphere(c, r, m)
    if lights_and_quads:
        for i in range(40):
            Q = ((rng.random(3) * 2 - 1) * np.array([25, 0, 25]) + np.array([0, 0.5 + rng.random() * 4, 0])).astype(np.float32)
            s.MakeQuad(Q, np.float32([rng.random() + 0.3, 0, 0]), np.float32([0, rng.random() + 0.3, rng.random()]), mats[int(rng.integers(0, len(mats)))])
        s.set_background((0.02, 0.03, 0.05))
    s.BuildBVH_TopDown()
    return s


@pytest.mark.parametrize("ext", [False, True])
def test_world_larger_than_the_lds_renders_bit_exact_from_global_memory(p, ext):
    """3000 spheres = 2999 wide nodes = 228 KB of node records alone: more than the 160 KiB LDS.  The renderer must keep
    the streaming kernel (not fall back to the wave-per-pixel baseline) and the framebuffer must still equal the oracle's."""
    s = _big_sphere_scene(p, 2999, 77, lights_and_quads=ext)
    W, H, spp = 120, 80, 6
    cam = p.DefocusBlurCamera((26, 4, 6), (0, 1, 0), (0, 1, 0), 35.0, W / H, 0.05, 24.0)
    w = s.getWorldPtr()
    assert w.n_nodes >= 5999
    for variant in (0, 2):
        r = p.Renderer.MakeRenderer(W, H, spp, 50, cam, w, variant=variant)
        info = r.kernel_info()
        assert info["variant"] == (3 if variant == 0 else 2) and not info["lds_resident"], info
        r.Render()
        img = r.DownloadRenderbuffer()
        r.close()
        ref, _ = O.render(as_oracle_world(w), as_oracle_camera(cam), W, H, spp, 50)
        assert bits_equal(img, ref), f"variant {variant}: " + mismatch_report(img, ref)
    with pytest.raises(p.capi.RtError, match="variant 4"):
        p.Renderer.MakeRenderer(W, H, spp, 50, cam, w, variant=4)


def _big_list_or_tree(p, kind, n, seed):
    """n random spheres as a HittableList (HittableList.cuh) or as a bvh_node tree built pairwise bottom-up (bvh_node.cuh)"""
    rng = np.random.default_rng(seed)
    s = p.Scene()
    mats = [s.Lambertian((0.7, 0.3, 0.3)), s.Metal((0.8, 0.8, 0.7), 0.2), s.Dielectric((1, 1, 1), 1.5), s.Lambertian((0.4, 0.6, 0.4))]
    refs, xs = [], []
    for i in range(n):
        c = ((rng.random(3) * 2 - 1) * np.array([30, 0, 30]) + np.array([0, 0.2 + rng.random() * 3, 0])).astype(np.float32)
        r = float(0.05 + rng.random() * 0.25)
        m = mats[int(rng.integers(0, len(mats)))]
        prim = s.MakeMovingSphere(c, c + np.float32([0, 0.3, 0]), r, m) if i % 7 == 0 else s.MakeSphere(c, r, m)
        refs.append(s.prim_ref(prim))
        xs.append(float(c[0]))
    if kind == "list":
        s.MakeHittableList()
        return s
    order = np.argsort(xs)   # neighbours in x pair up first: a usable tree
    refs = [refs[i] for i in order]
    while len(refs) > 1:
        nxt = [s.bvh_node(refs[i], refs[i + 1]) for i in range(0, len(refs) - 1, 2)]
        if len(refs) % 2:
            nxt.append(refs[-1])
        refs = nxt
    s.set_world_node_tree(refs[0])
    return s


@pytest.mark.parametrize("kind,n", [("list", 5500), ("tree", 3000)])
def test_list_and_tree_worlds_larger_than_the_lds_stay_on_the_streaming_kernel(p, kind, n):
    """A HittableList of 5500 spheres (176 KB of sphere records) and a bvh_node tree of 3000 (228 KB of nodes) exceed the 160 KiB
    LDS: they now run the global-memory form of the streaming kernel (HittableList.cuh:21-34 / bvh_node.cuh:19-24 order kept)
    instead of falling back to the wave-per-pixel baseline; framebuffer bit-identical to the oracle's."""
    s = _big_list_or_tree(p, kind, n, 31)
    W, H, spp = (48, 32, 2) if kind == "list" else (120, 80, 4)
    cam = p.DefocusBlurCamera((26, 4, 6), (0, 1, 0), (0, 1, 0), 35.0, W / H, 0.05, 24.0)
    w = s.getWorldPtr()
    r = p.Renderer.MakeRenderer(W, H, spp, 12, cam, w)
    info = r.kernel_info()
    assert info["variant"] == 2 and not info["lds_resident"], info
    r.Render()
    img = r.DownloadRenderbuffer()
    r.close()
    ref, _ = O.render(as_oracle_world(w), as_oracle_camera(cam), W, H, spp, 12)
    assert bits_equal(img, ref), mismatch_report(img, ref)
    base = p.Renderer.MakeRenderer(W, H, spp, 12, cam, w, variant=1)   # the former fallback: same image up to summation order
    base.Render()
    assert np.nanmax(np.abs(base.DownloadRenderbuffer() - ref)) < 1e-5
    base.close()


@pytest.mark.parametrize("which", ["three_spheres", "node_tree", "book2_moving"])
def test_forced_global_memory_path_agrees_for_every_world_kind(p, which, monkeypatch):
    """RT06_FORCE_BIG=1 on a HittableList, a bvh_node tree and a BVH that would fit the LDS; with RT06_FORCE_WIDE=1 also the 32-bit
    reference encoding (a BVH whose references fit 16 bits keeps them narrow on the global-memory path)"""
    W, H, spp = 96, 64, 6
    s = _node_tree_scene(p) if which == "node_tree" else config_scene(p, which)
    cam = config_cameras(p, "book2_moving" if which != "three_spheres" else which, W, H)
    images = []
    for env in ({}, {"RT06_FORCE_BIG": "1"}, {"RT06_FORCE_BIG": "1", "RT06_FORCE_WIDE": "1"}):
        for k in ("RT06_FORCE_BIG", "RT06_FORCE_WIDE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        r = p.Renderer.MakeRenderer(W, H, spp, 50, cam, s.getWorldPtr())
        assert r.kernel_info()["lds_resident"] == (not env)
        r.Render()
        images.append(r.DownloadRenderbuffer().tobytes())
        r.close()
    assert images[0] == images[1] == images[2]


def test_lds_and_global_memory_paths_agree_on_the_book_scene(p, monkeypatch):
    """RT06_FORCE_BIG=1 sends a world that WOULD fit the LDS down the global-memory path: same bits."""
    W, H, spp = 200, 120, 10
    s = config_scene(p, "book1_final")
    cam = config_cameras(p, "book1_final", W, H)
    r = p.Renderer.MakeRenderer(W, H, spp, 50, cam, s.getWorldPtr())
    assert r.kernel_info() == {"variant": 3, "lds_resident": True, "workgroup": 768, "workgroups_per_cu": 2}
    r.Render()
    a = r.DownloadRenderbuffer()
    r.close()
    monkeypatch.setenv("RT06_FORCE_BIG", "1")
    r = p.Renderer.MakeRenderer(W, H, spp, 50, cam, s.getWorldPtr())
    assert not r.kernel_info()["lds_resident"]
    r.Render()
    b = r.DownloadRenderbuffer()
    r.close()
    assert a.tobytes() == b.tobytes()

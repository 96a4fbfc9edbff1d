"""Frozen oracle outputs (tests/golden/frozen_*.npz, written by oracle/gen_frozen.py): SURVEY.md §8c G2-G7.

The live parity tests compare HIP with the oracle as both are TODAY; these fixtures hold yesterday's oracle, so a change that
moves oracle and kernels together is caught.  CPU tests: the oracle and the product's HOST code (scene generation, BVH
builder, camera constructors) reproduce the files bit for bit.  GPU tests: the device probes and the renderer do.
These are oracle outputs — what is pinned to the reference itself is glm_* / ref_* (tests/test_reference_pins.py).
"""
import ctypes as C
import os

import numpy as np
import pytest

import _oracle as O
from _common import bits_equal, mismatch_report, pkg

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SEED = 1984


def frozen(name):
    return np.load(os.path.join(GOLD, f"frozen_{name}.npz"))   # allow_pickle stays False: plain arrays only


def pod(cls, raw):
    c = cls()
    assert C.sizeof(c) == len(raw)
    C.memmove(C.byref(c), raw.tobytes(), len(raw))
    return c


CAMERA_ARGS = {"pinhole": ((0, 0, 0), (0, 0, -1), (0, 1, 0), 90.0, 400 / 225),
               "defocus": ((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, 1.5, 0.1, 10.0),
               "motion": ((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, 1.0, 0.0, 1.0)}


# ---- CPU: oracle ---------------------------------------------------------------------------------------------------------
def test_oracle_sphere_matches_frozen():
    f = frozen("sphere")
    t = np.zeros(len(f["t"]), np.float32)
    O.lib().orc_sphere_batch(len(t), f["rays"], f["spheres"], t)
    assert bits_equal(t, f["t"]), mismatch_report(t, f["t"])
    assert 0.2 < (f["t"] < 1e30).mean() < 0.9


@pytest.mark.parametrize("name", sorted(CAMERA_ARGS))
def test_oracle_and_product_camera_constructors_match_frozen(name):
    f = frozen("cameras")
    ctor = {"pinhole": O.camera_pinhole, "defocus": O.camera_defocus, "motion": O.camera_motion}[name]
    assert bytes(ctor(*CAMERA_ARGS[name])) == f[f"{name}_pod"].tobytes()
    p = pkg()
    pctor = {"pinhole": p.PinholeCamera, "defocus": p.DefocusBlurCamera, "motion": p.MotionBlurCamera}[name]
    assert bytes(pctor(*CAMERA_ARGS[name])) == f[f"{name}_pod"].tobytes()      # host code of librt06.so, no GPU needed


@pytest.mark.parametrize("name", sorted(CAMERA_ARGS))
def test_oracle_camera_rays_match_frozen(name):
    f = frozen("cameras")
    cam = pod(O.Camera, f[f"{name}_pod"])
    n = len(f["st"])
    rays, draws = np.zeros((n, 7), np.float32), np.zeros(n, np.uint32)
    O.lib().orc_camera_batch(SEED, C.byref(cam), n, f["st"], f["keys"], rays, draws)
    assert bits_equal(rays, f[f"{name}_rays"]) and np.array_equal(draws, f[f"{name}_draws"])


def test_oracle_scatter_matches_frozen():
    f = frozen("scatter")
    n = len(f["dist"])
    mats = np.ascontiguousarray(f["mats"])
    sc, rays, att, draws = np.zeros(n, np.int32), np.zeros((n, 7), np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.uint32)
    O.lib().orc_scatter_batch(SEED, n, mats.ctypes.data, f["rays"], f["dist"], f["normals"], f["keys"], sc, rays, att, draws)
    assert np.array_equal(sc, f["scattered"]) and np.array_equal(draws, f["draws"])
    assert bits_equal(rays, f["out_rays"]) and bits_equal(att, f["atten"])


@pytest.mark.parametrize("which", ["book1_final", "book2_moving"])
def test_oracle_and_product_scene_bytes_match_frozen(which):
    """host RNG stream + scene layout (Scenes.cu:219-270) + median-split builder (BVH.cu:180-210), oracle and librt06.so"""
    f = frozen(which)
    o = getattr(O.Scene, which)(SEED)
    assert o.nodes.tobytes() == f["nodes"].tobytes() and o.prims.tobytes() == f["prims"].tobytes() and o.materials.tobytes() == f["materials"].tobytes()
    assert o.world.root == int(f["root"])
    s = getattr(pkg().Scene, which)(SEED)
    nodes, prims, mats = s.arrays()
    assert nodes.tobytes() == f["nodes"].tobytes() and prims.tobytes() == f["prims"].tobytes() and mats.tobytes() == f["materials"].tobytes()
    assert s.getWorldPtr().root == int(f["root"])


@pytest.mark.parametrize("which", ["book1_final", "book2_moving"])
def test_oracle_trace_and_radiance_match_frozen(which):
    f = frozen(which)
    o = getattr(O.Scene, which)(SEED)
    n = len(f["t"])
    hit, t, prim, nrm = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros(n, np.int32), np.zeros((n, 3), np.float32)
    assert O.lib().orc_trace_batch(C.byref(o.world), n, f["rays"], hit, t, prim, nrm) == 0
    assert np.array_equal(hit, f["hit"]) and np.array_equal(prim, f["prim"]) and bits_equal(t, f["t"]) and bits_equal(nrm, f["normal"])
    cam = pod(O.Camera, f["camera_pod"])
    rad = np.zeros_like(f["radiance"])
    assert O.lib().orc_radiance_batch(C.byref(o.world), C.byref(cam), int(f["width"]), int(f["height"]), 50, SEED, len(rad), f["keys"], rad) == 0
    assert bits_equal(rad, f["radiance"]), mismatch_report(rad, f["radiance"])


def test_oracle_config1_image_matches_frozen():
    """BASELINE.json configs[0]: three spheres, 400 x 225, 1 spp, depth 50, on the CPU path"""
    f = frozen("config1_image")
    scene = O.Scene.three_spheres()   # keeps the arrays the world points to alive
    img, _ = O.render(scene.world, O.camera_pinhole(*CAMERA_ARGS["pinhole"]), 400, 225, 1, 50, SEED)
    assert bits_equal(img[..., :3], f["rgb"]) and bool(f["alpha_all_one"]) and np.all(img[..., 3] == 1.0)


# ---- GPU: HIP probes and renderer ------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_device_sphere_matches_frozen():
    f = frozen("sphere")
    t = pkg().api.probe_sphere(f["rays"], f["spheres"])
    assert bits_equal(t, f["t"]), mismatch_report(t, f["t"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CAMERA_ARGS))
def test_device_camera_rays_match_frozen(name):
    p = pkg()
    f = frozen("cameras")
    rays, draws = p.api.probe_camera(SEED, pod(p.capi.Camera, f[f"{name}_pod"]), f["st"], f["keys"])
    assert bits_equal(rays, f[f"{name}_rays"]) and np.array_equal(draws, f[f"{name}_draws"])


@pytest.mark.gpu
def test_device_scatter_matches_frozen():
    p = pkg()
    f = frozen("scatter")
    mats = np.frombuffer(np.ascontiguousarray(f["mats"]).tobytes(), dtype=p.capi.MAT_DT)
    sc, rays, att, draws = p.api.probe_scatter(SEED, mats, f["rays"], f["dist"], f["normals"], f["keys"])
    assert np.array_equal(sc, f["scattered"]) and np.array_equal(draws, f["draws"])
    assert bits_equal(rays, f["out_rays"]) and bits_equal(att, f["atten"])


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["book1_final", "book2_moving"])
def test_device_trace_and_radiance_match_frozen(which):
    p = pkg()
    f = frozen(which)
    s = getattr(p.Scene, which)(SEED)
    w = s.getWorldPtr()
    hit, t, prim, nrm = p.api.probe_trace(w, f["rays"])
    assert np.array_equal(hit, f["hit"]) and np.array_equal(prim, f["prim"]) and bits_equal(t, f["t"]) and bits_equal(nrm, f["normal"])
    cfg = p.capi.RenderConfig(int(f["width"]), int(f["height"]), 500, 50, SEED, 0, 0, 1, 0)
    rad = p.api.probe_radiance(cfg, pod(p.capi.Camera, f["camera_pod"]), w, f["keys"])
    assert bits_equal(rad, f["radiance"]), mismatch_report(rad, f["radiance"])


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1])
def test_device_config1_image_matches_frozen(variant):
    """the config-1 frame through Renderer.MakeRenderer / Render / DownloadRenderbuffer (HittableList world).  The streaming kernel
    (variant 0) is bit-exact; the wave-per-pixel baseline (variant 1) at 1 spp has a single term per pixel sum, so it is too."""
    p = pkg()
    f = frozen("config1_image")
    scene = p.Scene.three_spheres()
    r = p.Renderer.MakeRenderer(400, 225, 1, 50, p.PinholeCamera(*CAMERA_ARGS["pinhole"]), scene.getWorldPtr(), seed=SEED, variant=variant)
    r.Render()
    img = r.DownloadRenderbuffer()
    r.close()
    assert bits_equal(img[..., :3], f["rgb"]), mismatch_report(img[..., :3], f["rgb"])
    assert np.all(img[..., 3] == 1.0)

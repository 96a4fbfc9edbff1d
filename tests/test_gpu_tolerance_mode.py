"""Kernel variant 6, the opt-in TOLERANCE MODE of the streaming kernel: the twelve plane parameters of a node visit are (b - o) * RN(1/d) instead of
aabb.cuh:30-31's quotients (which the default kernel reproduces bit for bit).  Its contract is BASELINE.json's |delta| < 1e-3 per channel against
the CPU oracle, not the oracle's bits; the default (variant 0) never resolves to it and worlds beyond the reference's feature set (quads, lights, media) or beyond the LDS refuse it.  Round 4
measured it bit-identical on the full-size frames of BASELINE configs[1..2] (tools/tolerance_mode.py, EXPERIMENTS.md E4); here the tolerance is what is
asserted and the bit-identity is only reported."""
import numpy as np
import pytest

import _oracle as O
from _common import bits_equal, config_cameras, config_scene, pkg

pytestmark = pytest.mark.gpu


def _oracle_frame(which, W, H, spp, depth):
    scene, cam = config_scene(O, which), config_cameras(O_cams(), which, W, H)
    ref, _ = O.render(scene.world, cam, W, H, spp, depth)
    return ref


class _OCams:   # config_cameras() speaks the product's camera vocabulary; the oracle's constructors under those names
    PinholeCamera = staticmethod(O.camera_pinhole)
    DefocusBlurCamera = staticmethod(O.camera_defocus)
    MotionBlurCamera = staticmethod(O.camera_motion)


def O_cams():
    return _OCams


@pytest.mark.parametrize("which,W,H,spp", [("book1_final", 300, 200, 16), ("book2_moving", 200, 200, 16)])
def test_tolerance_mode_is_inside_the_stated_tolerance(which, W, H, spp):
    p = pkg()
    scene, cam = config_scene(p, which), config_cameras(p, which, W, H)
    r = p.Renderer.MakeRenderer(W, H, spp, 50, cam, scene.getWorldPtr(), variant=6)
    assert r.kernel_info()["variant"] == 6 and r.kernel_info()["lds_resident"]
    r.Render()
    img = r.DownloadRenderbuffer()
    r.close()
    ref = _oracle_frame(which, W, H, spp, 50)
    assert np.array_equal(np.isnan(img), np.isnan(ref))
    d = np.abs(img - ref)
    d[np.isnan(d)] = 0.0
    print(f"{which}: bit-identical {bits_equal(img, ref)}, max |delta| {d.max():.3e}")
    assert d.max() < 1e-3   # north_star's per-channel tolerance


def test_the_default_never_resolves_to_tolerance_mode_and_other_worlds_refuse_it():
    p = pkg()
    scene, cam = config_scene(p, "book1_final"), config_cameras(p, "book1_final", 64, 48)
    r = p.Renderer.MakeRenderer(64, 48, 2, 8, cam, scene.getWorldPtr())
    assert r.kernel_info()["variant"] == 3
    r.close()
    big = config_scene(p, "book2_final")   # records in global memory: not instantiated (measured +8 %, below the acceptance bar)
    with pytest.raises(p.capi.RtError):
        p.Renderer.MakeRenderer(64, 48, 2, 8, config_cameras(p, "book2_final", 64, 48), big.getWorldPtr(), variant=6)
    quads = config_scene(p, "cornell_box")   # quads: a quad's edges are its box's edges — the Cornell box at its own 5000 spp left the tolerance in one pixel
    with pytest.raises(p.capi.RtError):
        p.Renderer.MakeRenderer(64, 48, 2, 8, config_cameras(p, "cornell_box", 64, 48), quads.getWorldPtr(), variant=6)
    with pytest.raises(p.capi.RtError):
        p.Renderer.MakeRenderer(64, 48, 2, 8, cam, scene.getWorldPtr(), variant=7)


def test_cpp_mirror_can_opt_into_tolerance_mode(tmp_path):
    """Renderer::MakeRenderer(..., variant = Renderer::kToleranceMode) of include/rt06/rt06.hpp, through tests/cpp/first_app (the reference's FirstApp flow:
    live Book-2 moving scene, MotionBlurCamera vfov 30, shutter 0.1..1): inside the tolerance against the oracle's frame"""
    import os
    import subprocess
    from _common import ROOT
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")])
    W, H, spp, depth = 320, 180, 8, 50
    raw = str(tmp_path / "frame.f32")
    subprocess.check_call([os.path.join(ROOT, "tests", "cpp", "first_app"), "render", str(W), str(H), str(spp), str(depth), raw, "--variant", "6"], stdout=subprocess.DEVNULL)
    got = np.fromfile(raw, dtype=np.float32).reshape(H, W, 4)
    scene = O.Scene.book2_moving(1984)
    ref, _ = O.render(scene.world, O.camera_motion((13, 2, 3), (0, 0, 0), (0, 1, 0), 30.0, W / H, 0.1, 1.0), W, H, spp, depth)
    d = np.abs(got - ref)
    d[np.isnan(d)] = 0.0
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and d.max() < 1e-3

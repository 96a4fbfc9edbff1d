"""render_kernel_xchg (kernel variant 5): the streaming kernel with rays exchanged between tracer and shader waves through LDS rings.
Same lane program as variant 3, other schedule: every frame must be the oracle's and variant 3's bit for bit, for every setting of
the roles / ring thresholds, tiny and ragged frames, depth limits, moving spheres, rays outside the fast-division class, sharding
and several passes.  (tests/test_gpu_parity.py also runs its variant-parametrised tests with variant 5.)"""
import numpy as np
import pytest

import _oracle as O
from _common import as_oracle_camera, as_oracle_world, bits_equal, config_cameras, config_scene, mismatch_report, pkg

pytestmark = pytest.mark.gpu


def _gpu(p, scene, cam, W, H, spp, depth=50, variant=5, **kw):
    r = p.Renderer.MakeRenderer(W, H, spp, depth, cam, scene.getWorldPtr(), variant=variant, **kw)
    r.Render()
    img = r.DownloadRenderbuffer()
    info = r.kernel_info()
    r.close()
    return img, info


def _cpu(scene, cam, W, H, spp, depth=50):
    ref, _ = O.render(as_oracle_world(scene.getWorldPtr()), as_oracle_camera(cam), W, H, spp, depth)
    return ref


@pytest.mark.parametrize("which,W,H,spp,depth", [
    ("book1_final", 96, 64, 8, 50), ("book1_final", 8, 8, 1, 50), ("book1_final", 1, 1, 1, 50), ("book1_final", 13, 7, 3, 50),
    ("book1_final", 203, 117, 19, 50), ("book1_final", 64, 64, 9, 1), ("book1_final", 64, 64, 9, 2), ("book1_final", 64, 48, 5, 0),
    ("book2_moving", 160, 120, 12, 50), ("book2_moving", 33, 21, 40, 50), ("book1_final", 400, 300, 24, 50),
])
def test_exchange_kernel_frame_is_the_oracles(which, W, H, spp, depth):
    p = pkg()
    scene, cam = config_scene(p, which), config_cameras(p, which, W, H)
    img, info = _gpu(p, scene, cam, W, H, spp, depth)
    assert info["variant"] == 5 and info["lds_resident"]
    ref = _cpu(scene, cam, W, H, spp, depth)
    assert bits_equal(img, ref), mismatch_report(img, ref)


@pytest.mark.parametrize("setting", ["4,64,8,16,2,1,40,1", "10,256,32,64,10,0,56,2", "8,0,1,1,0,1,64,1", "11,192,64,64,50,1,1,1", "1,128,16,48,6,1,44,1", "6,96,24,32,3,0,30,2",
                                     "8,192,16,48,6,1,44,2", "9,64,16,48,6,1,44,2", "3,16,4,8,1,1,20,2"])
def test_exchange_kernel_schedule_settings_do_not_change_a_bit(monkeypatch, setting):
    """RT06_XCHG = tracer waves, rays beyond the tracer lanes, exchange threshold, shade threshold, patience, shader priority, hot-loop keep, ring pairs"""
    p = pkg()
    W, H, spp = 240, 160, 20
    scene, cam = config_scene(p, "book1_final"), config_cameras(p, "book1_final", W, H)
    ref, _ = _gpu(p, scene, cam, W, H, spp, variant=3)
    monkeypatch.setenv("RT06_XCHG", setting)
    img, info = _gpu(p, scene, cam, W, H, spp)
    assert info["variant"] == 5
    assert img.tobytes() == ref.tobytes()


def test_exchange_kernel_equals_the_streaming_kernel_on_a_big_frame_and_is_repeatable():
    p = pkg()
    W, H, spp = 1200, 800, 16
    scene, cam = config_scene(p, "book1_final"), config_cameras(p, "book1_final", W, H)
    a, _ = _gpu(p, scene, cam, W, H, spp, variant=3)
    b, _ = _gpu(p, scene, cam, W, H, spp)
    c, _ = _gpu(p, scene, cam, W, H, spp)
    assert a.tobytes() == b.tobytes() == c.tobytes()


def test_exchange_kernel_rays_outside_the_fast_division_class():
    """a camera looking exactly down an axis: primary rays with zero direction components take the marked-reference loop"""
    p = pkg()
    W, H, spp = 65, 65, 6
    scene = config_scene(p, "book1_final")
    cam = p.PinholeCamera((0, 1, 8), (0, 1, 0), (0, 1, 0), 40.0, W / H)
    img, _ = _gpu(p, scene, cam, W, H, spp)
    ref = _cpu(scene, cam, W, H, spp)
    assert bits_equal(img, ref), mismatch_report(img, ref)


def test_exchange_kernel_small_worlds():
    """a one-sphere world (the root reference is a leaf) and a two-sphere world, rays that miss the world box at once"""
    p = pkg()
    for n in (1, 2, 5):
        s = p.Scene()
        m = s.Lambertian((0.5, 0.6, 0.7))
        g = s.Metal((0.9, 0.9, 0.9), 0.1)
        for k in range(n):
            s.MakeSphere((k * 1.5 - 1.0, 0.25 * k, -3.0 - k), 0.7, m if k % 2 == 0 else g)
        s.BuildBVH_TopDown()
        W, H, spp = 72, 40, 7
        cam = p.PinholeCamera((0, 0, 0), (0, 0, -1), (0, 1, 0), 60.0, W / H)
        img, info = _gpu(p, s, cam, W, H, spp)
        assert info["variant"] == 5
        ref = _cpu(s, cam, W, H, spp)
        assert bits_equal(img, ref), (n, mismatch_report(img, ref))


def test_exchange_kernel_sharded_and_multi_pass(monkeypatch):
    import torch
    p = pkg()
    W, H, spp = 203, 117, 11
    scene, cam = config_scene(p, "book2_moving"), config_cameras(p, "book2_moving", W, H)
    ref, _ = _gpu(p, scene, cam, W, H, spp, variant=3)
    monkeypatch.setenv("RT06_PASS_SPP", "4")
    many, _ = _gpu(p, scene, cam, W, H, spp)
    assert many.tobytes() == ref.tobytes()
    shards, last = [], None
    for rank in range(3):
        r = p.Renderer.MakeRenderer(W, H, spp, 50, cam, scene.getWorldPtr(), rank=rank, world_size=3, variant=5)
        buf = torch.zeros(r.shard_floats(), dtype=torch.float32, device="cuda:0")
        r.render_async(torch.cuda.current_stream().cuda_stream, buf.data_ptr())
        torch.cuda.synchronize()
        shards.append(buf)
        if last is not None:
            last.close()
        last = r
    image = torch.empty(H * W * 4, dtype=torch.float32, device="cuda:0")
    last.assemble(torch.cat(shards).data_ptr(), image.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    last.close()
    assert image.cpu().numpy().tobytes() == ref.tobytes()


def test_exchange_kernel_refuses_worlds_it_cannot_take():
    p = pkg()
    cb, cam = config_scene(p, "cornell_box"), config_cameras(p, "cornell_box", 64, 64)
    with pytest.raises(p.capi.RtError, match="variant 5"):
        p.Renderer.MakeRenderer(64, 64, 1, 5, cam, cb.getWorldPtr(), variant=5)
    ts, cam = config_scene(p, "three_spheres"), config_cameras(p, "three_spheres", 64, 36)
    with pytest.raises(p.capi.RtError, match="variant 5"):
        p.Renderer.MakeRenderer(64, 36, 1, 5, cam, ts.getWorldPtr(), variant=5)

"""BASELINE.json configs[3] and configs[4] at the geometry BASELINE states — 600x600 tile-sharded on 4 GPUs, 3840x2160 on 8 GPUs
(480 x 270 tiles, several passes of the per-pass sample buffer) — on the one GPU the test box has: every rank's shard is rendered
there one after another, concatenated rank-major as the frame-end gather would, assembled, and compared with the single-GPU frame
bit for bit; the frame itself is checked against the CPU oracle on a sparse set of pixels (O.render_pixels).
The spp is reduced (the sample loop is the same code for 16 or 10000 spp; several passes are forced through the pass budget);
both scenes are extensions beyond the reference (parity unpinned by construction: tests/test_gpu_cornell.py)."""
import numpy as np
import pytest

import _oracle as O
from _common import as_oracle_camera, as_oracle_world, bits_equal, config_cameras, config_scene, mismatch_report, pkg

pytestmark = pytest.mark.gpu


def _single(p, W, H, spp, depth, cam, w):
    r = p.Renderer.MakeRenderer(W, H, spp, depth, cam, w)
    r.Render()
    img = r.DownloadRenderbuffer()
    info = r.kernel_info()
    r.close()
    return img, info


def _sharded(p, W, H, spp, depth, cam, w, world_size):
    import torch
    shards, last = [], None
    for rank in range(world_size):
        r = p.Renderer.MakeRenderer(W, H, spp, depth, cam, w, rank=rank, world_size=world_size)
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
    return image.cpu().numpy().reshape(H, W, 4)


def _sparse_check(img, w, cam, W, H, spp, depth, n, seed):
    rng = np.random.default_rng(seed)
    gids = rng.integers(0, W * H, n).astype(np.uint32)
    gids[:4] = (0, W - 1, (H - 1) * W, H * W - 1)   # the four corners: first / last tile of the first / last tile row
    exp = O.render_pixels(as_oracle_world(w), as_oracle_camera(cam), W, H, spp, depth, gids)
    got = img.reshape(-1, 4)[gids]
    assert bits_equal(got, exp), mismatch_report(got, exp)


def test_config5_geometry_3840x2160_sparse_parity_and_multi_pass(monkeypatch):
    """the Book-2 final scene at 3840 x 2160 (480 x 270 tiles), depth 50; 6 spp in 3 passes of 2; 32 pixels against the oracle"""
    p = pkg()
    W, H, spp, depth = 3840, 2160, 6, 50
    scene, cam = config_scene(p, "book2_final"), config_cameras(p, "book2_final", W, H)
    w = scene.getWorldPtr()
    monkeypatch.setenv("RT06_PASS_SPP", "2")   # two samples per pixel per pass
    img, info = _single(p, W, H, spp, depth, cam, w)
    assert info["variant"] == 3 and not info["lds_resident"]             # the global-memory form of the streaming kernel
    assert np.isfinite(img).all() and np.all(img[..., 3] == 1.0)
    _sparse_check(img, w, cam, W, H, spp, depth, 32, 50)
    monkeypatch.delenv("RT06_PASS_SPP")
    one_pass, _ = _single(p, W, H, spp, depth, cam, w)
    assert one_pass.tobytes() == img.tobytes()


def test_config5_at_its_own_resolution_with_natural_passes():
    """configs[4] at 3840 x 2160, depth 50, with the REAL pass budget (120 GiB of per-sample buffers, or 45 % of the free HBM = up to 258 spp per
    pass at this size): 800 spp = 4 natural passes (258 + 258 + 258 + 26), i.e. the regime of the 10 000-spp run (39 passes) — running sums
    carried from pass to pass, the work counter reset per pass, the per-pass buffers at their full size (129 GB).  32 random pixels + the four
    corners against the CPU oracle, bit for bit."""
    p = pkg()
    W, H, spp, depth = 3840, 2160, 800, 50
    scene, cam = config_scene(p, "book2_final"), config_cameras(p, "book2_final", W, H)
    w = scene.getWorldPtr()
    r = p.Renderer.MakeRenderer(W, H, spp, depth, cam, w)
    pi = r.pass_info()
    assert pi["n_passes"] >= 3 and pi["pass_spp"] * pi["n_passes"] >= spp, pi
    assert pi["buffer_bytes"] <= (120 << 30) + W * H * 16, pi            # the budget covers EVERY per-sample buffer of the pass
    assert pi["pass_spp"] * W * H * pi["bytes_per_sample"] <= 120 << 30, pi
    r.Render()
    img = r.DownloadRenderbuffer()
    times = r.kernel_times()
    total = r.last_kernel_ms()
    r.close()
    assert np.isfinite(img).all() and np.all(img[..., 3] == 1.0)
    assert 0.5 * total < sum(times) <= 1.02 * total, (times, total)    # the per-kernel times cover all passes, not the last one
    _sparse_check(img, w, cam, W, H, spp, depth, 36, 51)


def test_natural_and_forced_pass_cuts_give_the_same_frame(monkeypatch):
    """the same 960 x 540 x 48 frame of the Book-2 final scene in one pass, in 3 passes of 16 and in 7 passes of 7 (the last one short)"""
    p = pkg()
    W, H, spp, depth = 960, 540, 48, 50
    scene, cam = config_scene(p, "book2_final"), config_cameras(p, "book2_final", W, H)
    w = scene.getWorldPtr()
    one, _ = _single(p, W, H, spp, depth, cam, w)
    for per_pass in (16, 7):
        monkeypatch.setenv("RT06_PASS_SPP", str(per_pass))
        many, _ = _single(p, W, H, spp, depth, cam, w)
        assert many.tobytes() == one.tobytes(), per_pass


def test_pass_budget_counts_every_per_sample_buffer(monkeypatch):
    """RT06_PASS_BUDGET_BYTES bounds sample buffer + primary-ray records together (it used to count 12 of 60 bytes)"""
    p = pkg()
    W, H, spp, depth = 640, 360, 40, 8
    scene, cam = config_scene(p, "book1_final"), config_cameras(p, "book1_final", W, H)
    budget = 100 << 20
    monkeypatch.setenv("RT06_PASS_BUDGET_BYTES", str(budget))
    r = p.Renderer.MakeRenderer(W, H, spp, depth, cam, scene.getWorldPtr())
    pi = r.pass_info()
    r.close()
    n_px = ((W + 7) // 8) * ((H + 7) // 8) * 64
    assert pi["n_passes"] > 1 and pi["pass_spp"] * n_px * pi["bytes_per_sample"] <= budget < (pi["pass_spp"] + 1) * n_px * pi["bytes_per_sample"], pi
    assert pi["buffer_bytes"] == pi["pass_spp"] * n_px * pi["bytes_per_sample"] + n_px * 16, pi


def test_kernel_times_sum_over_the_passes_of_a_render(monkeypatch):
    """rt_renderer_kernel_times of a forced 3-pass render is about three times that of one pass of a third of the samples"""
    p = pkg()
    W, H, depth = 1200, 800, 50
    scene, cam = config_scene(p, "book1_final"), config_cameras(p, "book1_final", W, H)
    w = scene.getWorldPtr()

    def stream_ms(spp):
        r = p.Renderer.MakeRenderer(W, H, spp, depth, cam, w)
        n = r.pass_info()["n_passes"]
        r.Render(); r.Render()
        t = r.kernel_times()
        tot = r.last_kernel_ms()
        r.close()
        return n, t, tot
    n1, t1, _ = stream_ms(32)
    monkeypatch.setenv("RT06_PASS_SPP", "32")
    n3, t3, tot3 = stream_ms(96)
    assert (n1, n3) == (1, 3)
    assert 2.2 * t1[1] < t3[1] < 3.8 * t1[1], (t1, t3)
    assert sum(t3) <= 1.02 * tot3


@pytest.mark.parametrize("world_size", [4, 8])
def test_config5_geometry_sharded_over_4_and_8_ranks_is_the_same_frame(world_size):
    p = pkg()
    W, H, spp, depth = 3840, 2160, 2, 50
    scene, cam = config_scene(p, "book2_final"), config_cameras(p, "book2_final", W, H)
    w = scene.getWorldPtr()
    ref, _ = _single(p, W, H, spp, depth, cam, w)
    got = _sharded(p, W, H, spp, depth, cam, w, world_size)
    assert got.tobytes() == ref.tobytes()


def test_config4_cornell_600x600_sharded_over_4_ranks_is_the_same_frame():
    """configs[3] as BASELINE states it: 600 x 600, tile-sharded on 4 GPUs; 64 spp here instead of 5000"""
    p = pkg()
    W, H, spp, depth = 600, 600, 64, 50
    scene, cam = config_scene(p, "cornell_box"), config_cameras(p, "cornell_box", W, H)
    w = scene.getWorldPtr()
    ref, info = _single(p, W, H, spp, depth, cam, w)
    assert info["lds_resident"]
    got = _sharded(p, W, H, spp, depth, cam, w, 4)
    assert got.tobytes() == ref.tobytes()
    _sparse_check(ref, w, cam, W, H, spp, depth, 48, 44)


def test_passes_shrink_when_the_device_cannot_hold_them(monkeypatch):
    """a budget the device does not have (400 GiB of per-sample buffers on a 288-GB GPU) is not an error: the passes are halved until their
    buffers fit (creation used to fail with hipMalloc's error instead of falling back to more passes)"""
    import torch
    p = pkg()
    W, H, spp, depth = 3840, 2160, 5000, 50
    scene, cam = config_scene(p, "book2_final"), config_cameras(p, "book2_final", W, H)
    monkeypatch.setenv("RT06_PASS_BUDGET_BYTES", str(400 << 30))
    asked = (400 << 30) // (W * H * 60)
    r = p.Renderer.MakeRenderer(W, H, spp, depth, cam, scene.getWorldPtr())
    pi = r.pass_info()
    r.close()
    total = torch.cuda.get_device_properties(0).total_memory
    assert asked * W * H * 60 > total                      # what was asked for cannot exist on this device
    assert pi["pass_spp"] < asked and pi["buffer_bytes"] < total and pi["pass_spp"] * pi["n_passes"] >= spp, pi

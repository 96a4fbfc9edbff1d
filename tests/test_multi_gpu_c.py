"""The multi-GPU renderer behind the C ABI (rt_multi_renderer_*: one host process, N GPUs, one grouped RCCL exchange).

CPU: the C-side shard layout is the one ray-tracing-v06_amd/multigpu.py and the gloo tests use, and the per-rank pixel maps
partition the frame.  GPU (one-GPU box): the N = 1 communicator (ncclCommInitAll over one device, self send/recv inside a group)
renders the same bits as the plain renderer, and bad device lists are refused.  N > 1 with RCCL needs N GPUs (RCCL refuses two
ranks on one device), so that exchange is exercised by the driver's scaling run only — but the whole N > 1 BRANCH of
rt_multi_renderer_* (per-rank renders side by side, shard offsets, cross-stream ordering, assemble_kernel, download) runs here
with RT06_MULTI_TRANSPORT=memcpy, which lets 2 / 4 / 8 ranks share device 0 and moves the shards with hipMemcpyAsync on the ranks'
own streams.  Layout and assembly are also covered by tests/test_dist_gloo.py (gloo, 2 and 3 processes) and by
test_tile_sharding_is_gpu_count_invariant (N = 2, 3, 8 on one GPU)."""
import numpy as np
import pytest

from _common import config_cameras, config_scene, pkg


@pytest.mark.parametrize("W,H,N", [(1200, 800, 8), (1200, 800, 1), (600, 600, 4), (3840, 2160, 8), (203, 117, 3), (7, 5, 2), (64, 8, 16)])
def test_c_shard_layout_equals_python_tile_layout_and_partitions_the_frame(W, H, N):
    p = pkg()
    from ray_tracing_v06_amd import multigpu
    assert p.api.shard_layout(W, H, N) == multigpu.tile_layout(W, H, N)
    tiles_x, n_tiles, n_local, shard_floats = p.api.shard_layout(W, H, N)
    seen = np.zeros(W * H, np.int32)
    for rank in range(N):
        gid = p.api.shard_pixel_map(W, H, N, rank)
        assert len(gid) == n_local * 64 == shard_floats // 4
        real = gid[gid != 0xFFFFFFFF]
        np.add.at(seen, real, 1)
        # independent statement of the mapping: shard position L -> tile L // 64 of this rank -> global tile t = tile * N + rank
        L = np.nonzero(gid != 0xFFFFFFFF)[0]
        t = (L // 64) * N + rank
        x = (t % tiles_x) * 8 + (L % 64) % 8
        y = (t // tiles_x) * 8 + (L % 64) // 8
        assert np.array_equal(real, (y * W + x).astype(np.uint32))
    assert np.all(seen == 1), "every pixel belongs to exactly one rank"


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["book1_final", "cornell_box"])
def test_multi_renderer_with_one_gpu_is_the_plain_renderer(which):
    """N = 1: a real RCCL communicator and a real (self) exchange of the frame on devices[0]"""
    p = pkg()
    W, H, spp, depth = 203, 117, 5, 12
    scene, cam = config_scene(p, which), config_cameras(p, which, W, H)
    w = scene.getWorldPtr()
    r = p.Renderer.MakeRenderer(W, H, spp, depth, cam, w)
    r.Render()
    ref = r.DownloadRenderbuffer()
    r.close()
    m = p.MultiRenderer.MakeRenderer(W, H, spp, depth, cam, w, 1)
    for _ in range(2):   # twice: the communicator and buffers are reused across frames
        m.Render()
        img = m.DownloadRenderbuffer()
        assert img.tobytes() == ref.tobytes()
    total, kernels, exchange = m.times()
    assert total > 0 and kernels > 0 and exchange >= 0 and total >= kernels * 0.5
    m.close()


@pytest.mark.gpu
def test_multi_renderer_refuses_bad_device_lists():
    p = pkg()
    scene, cam = config_scene(p, "three_spheres"), config_cameras(p, "three_spheres", 64, 36)
    w = scene.getWorldPtr()
    n_dev = p.api.device_count()
    for n, devs in ((0, None), (n_dev + 1, None), (1, [n_dev]), (1, [-1])):
        with pytest.raises(p.capi.RtError) as e:
            p.MultiRenderer.MakeRenderer(64, 36, 1, 4, cam, w, n, devices=devs)
        assert e.value.code == 1   # RT_ERR_INVALID
    if n_dev >= 2:
        with pytest.raises(p.capi.RtError):
            p.MultiRenderer.MakeRenderer(64, 36, 1, 4, cam, w, 2, devices=[0, 0])


@pytest.mark.gpu
def test_multi_renderer_over_all_gpus_of_the_box_matches_single_gpu():
    """runs the real N > 1 exchange wherever more than one GPU is visible (skipped on a one-GPU box)"""
    p = pkg()
    n_dev = p.api.device_count()
    if n_dev < 2:
        pytest.skip("one GPU visible: RCCL refuses two ranks on one device")
    W, H, spp, depth = 600, 400, 8, 50
    scene, cam = config_scene(p, "book1_final"), config_cameras(p, "book1_final", W, H)
    w = scene.getWorldPtr()
    r = p.Renderer.MakeRenderer(W, H, spp, depth, cam, w)
    r.Render()
    ref = r.DownloadRenderbuffer()
    r.close()
    m = p.MultiRenderer.MakeRenderer(W, H, spp, depth, cam, w, min(n_dev, 6))
    m.Render()
    assert m.DownloadRenderbuffer().tobytes() == ref.tobytes()
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("which,W,H,spp,n", [("book1_final", 1200, 800, 6, 2), ("book1_final", 1200, 800, 6, 4), ("cornell_box", 600, 600, 8, 4),
                                              ("book2_moving", 203, 117, 5, 3), ("book1_final", 1200, 800, 4, 8)])
def test_multi_renderer_n_ranks_on_one_gpu_with_the_memcpy_transport(monkeypatch, which, W, H, spp, n):
    """the N > 1 branch of rt_multi_renderer_render on ONE GPU: N ranks on device 0, shards moved by hipMemcpyAsync instead of
    ncclSend / ncclRecv; the assembled frame is the single-GPU frame bit for bit, twice in a row (buffers and events are reused)"""
    p = pkg()
    depth = 50
    scene, cam = config_scene(p, which), config_cameras(p, which, W, H)
    w = scene.getWorldPtr()
    r = p.Renderer.MakeRenderer(W, H, spp, depth, cam, w)
    r.Render()
    ref = r.DownloadRenderbuffer()
    r.close()
    monkeypatch.setenv("RT06_MULTI_TRANSPORT", "memcpy")
    m = p.MultiRenderer.MakeRenderer(W, H, spp, depth, cam, w, n)
    for _ in range(2):
        m.Render()
        img = m.DownloadRenderbuffer()
        assert img.tobytes() == ref.tobytes()
    total, kernels, exchange = m.times()
    assert total > 0 and kernels > 0 and 0 <= exchange < total
    m.close()


@pytest.mark.gpu
def test_multi_renderer_memcpy_transport_multi_pass_and_explicit_devices(monkeypatch):
    """several passes per rank (running sums) under the multi-renderer, device list given explicitly with repeats"""
    p = pkg()
    W, H, spp, depth = 320, 200, 12, 20
    scene, cam = config_scene(p, "book2_moving"), config_cameras(p, "book2_moving", W, H)
    w = scene.getWorldPtr()
    r = p.Renderer.MakeRenderer(W, H, spp, depth, cam, w)
    r.Render()
    ref = r.DownloadRenderbuffer()
    r.close()
    monkeypatch.setenv("RT06_MULTI_TRANSPORT", "memcpy")
    monkeypatch.setenv("RT06_PASS_SPP", "5")
    m = p.MultiRenderer.MakeRenderer(W, H, spp, depth, cam, w, 3, devices=[0, 0, 0])
    m.Render()
    assert m.DownloadRenderbuffer().tobytes() == ref.tobytes()
    m.close()
    monkeypatch.setenv("RT06_MULTI_TRANSPORT", "carrier-pigeon")
    with pytest.raises(p.capi.RtError):
        p.MultiRenderer.MakeRenderer(W, H, spp, depth, cam, w, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("which,traversal", [("book2_final", 0), ("book1_final", 1), ("cornell_box", 1), ("book2_moving", 2)])
def test_multi_renderer_memcpy_transport_on_the_global_memory_and_queue_kernels(monkeypatch, which, traversal):
    """the same N > 1 branch over the other kernel families: the global-memory form (Book-2 final scene) and the queue mode"""
    p = pkg()
    W, H, spp, depth = 160, 96, 6, 50
    scene, cam = config_scene(p, which), config_cameras(p, which, W, H)
    if traversal:
        scene.set_traversal(traversal)
    w = scene.getWorldPtr()
    r = p.Renderer.MakeRenderer(W, H, spp, depth, cam, w)
    r.Render()
    ref = r.DownloadRenderbuffer()
    r.close()
    monkeypatch.setenv("RT06_MULTI_TRANSPORT", "memcpy")
    m = p.MultiRenderer.MakeRenderer(W, H, spp, depth, cam, w, 3)
    m.Render()
    assert m.DownloadRenderbuffer().tobytes() == ref.tobytes()
    m.close()

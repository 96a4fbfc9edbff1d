"""The reference's disabled distance-sorted traversal queue (`_USE_PRIO_QUEUE`, rt_engine/geometry/BVH.cu:17-49, :80-86) as an
opt-in traversal of BVH worlds (rt_scene_set_traversal / rt_world_flat.traversal = RT_TRAVERSAL_QUEUE), with the reference's
off-by-one (`distances[head]` written after `head++`, :37-39) fixed in oracle and kernels alike.  Best-first instead of
depth-first: on the Book-1 final scene it saves 0.6 % of the box tests and costs 4.8 % more sphere tests (first test below), which
is why it exists on the baseline kernel and the probes only.  Parity: GPU against the oracle, bit for bit; nothing of the
reference executes here (BVH.cu does not build in this image), so like the live traversal it is parity-unpinned.

Round 3 adds a second alternative rule over the same tree, RT_TRAVERSAL_WIDE4 (mode 2; SURVEY §8f rank 4 "wider nodes" — not in the reference, whose
nodes are binary): a visit tests the up-to-four grandchild boxes, nearest first.  Same oracle twin / same bit-exact bar / same kernels as the queue."""
import ctypes as C

import numpy as np
import pytest

import _oracle as O
from _common import as_oracle_camera, as_oracle_world, bits_equal, config_cameras, config_scene, mismatch_report, pkg, random_rays


def _trace_oracle(world, rays):
    n = len(rays)
    hit, t, prim, nrm = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros(n, np.int32), np.zeros((n, 3), np.float32)
    rc = O.lib().orc_trace_batch(C.byref(world), n, rays, hit, t, prim, nrm)
    return rc, hit, t, prim, nrm


def test_queue_traversal_finds_the_same_hits_with_slightly_fewer_box_tests():
    s = O.Scene.book1_final(1984)
    cam = O.camera_defocus((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, 1.5, 0.1, 10.0)
    W, H, spp = 300, 200, 4
    s.world.traversal = 0
    ref, c0 = O.render(s.world, cam, W, H, spp, 50)
    s.world.traversal = 1
    img, c1 = O.render(s.world, cam, W, H, spp, 50)
    assert np.nanmax(np.abs(img - ref)) < 1e-3                      # the same closest hits (rounding near-ties aside)
    assert 0.97 * c0.box_tests < c1.box_tests <= c0.box_tests        # best-first saves little: near-first already orders each node's children
    assert c1.leaf_tests >= c0.leaf_tests
    assert c1.max_stack <= 32 and c1.max_stack >= c0.max_stack       # the frontier of a best-first walk is wider than a depth-first stack
    print(f"box tests per sample {c0.box_tests / c0.samples:.2f} -> {c1.box_tests / c1.samples:.2f}, "
          f"leaf tests {c0.leaf_tests / c0.samples:.3f} -> {c1.leaf_tests / c1.samples:.3f}, frontier {c0.max_stack} -> {c1.max_stack}")


def _onion_world(cls, node_dt, prim_dt, mat_dt, levels=6):
    """a complete binary tree whose level-k boxes all begin at x = k: a ray along +x dequeues whole levels before any leaf, so the
    frontier of the queue reaches 2^levels = 64 entries"""
    n_leaves = 1 << levels
    n_nodes = 2 * n_leaves - 1
    nodes = np.zeros(n_nodes, dtype=node_dt)
    prims = np.zeros(n_leaves, dtype=prim_dt)
    mats = np.zeros(1, dtype=mat_dt)
    mats["albedo"] = 0.5
    for i in range(n_nodes):   # heap order: children of i are 2i+1, 2i+2; level = floor(log2(i+1))
        level = int(np.floor(np.log2(i + 1)))
        nodes[i]["min"], nodes[i]["max"] = (float(level), -1.0, -1.0), (100.0, 1.0, 1.0)
        if level == levels:
            k = i - (n_leaves - 1)
            nodes[i]["left"], nodes[i]["right"] = -1, k
            prims[k]["c0"], prims[k]["radius"], prims[k]["c1"], prims[k]["mat"] = (50.0 + k * 0.01, 0.0, 0.0), 0.5, (0, 0, 0), 0
        else:
            nodes[i]["left"], nodes[i]["right"] = 2 * i + 1, 2 * i + 2
    w = cls()
    w.kind, w.root, w.n_nodes, w.n_prims, w.n_materials, w.max_stack = 0, 0, n_nodes, n_leaves, 1, levels + 1
    for k in range(3):
        w.bounds_min[k], w.bounds_max[k] = float(nodes[0]["min"][k]), float(nodes[0]["max"][k])
    w.nodes, w.prims, w.materials = nodes.ctypes.data, prims.ctypes.data, mats.ctypes.data
    w.traversal = 1
    return w, (nodes, prims, mats)


def test_queue_overflow_is_reported_by_the_oracle():
    w, keep = _onion_world(O.World, O.NODE_DT, O.PRIM_DT, O.MAT_DT)
    rays = np.float32([[-10, 0, 0, 1, 0, 0, 0]])
    rc, *_ = _trace_oracle(w, rays)
    assert rc == 4
    w.traversal = 0
    rc, hit, t, prim, _ = _trace_oracle(w, rays)
    assert rc == 0 and hit[0] == 1 and prim[0] == 0


def test_traversal_mode_is_validated_on_the_host():
    p = pkg()
    s = p.Scene.three_spheres()   # a HittableList: the queue belongs to BVH worlds
    s.set_traversal(1)
    assert s.getWorldPtr().traversal == 0
    s.set_traversal(2)
    assert s.getWorldPtr().traversal == 0
    with pytest.raises(p.capi.RtError):
        s.set_traversal(3)
    b = p.Scene.book1_final(1984).set_traversal(1)
    assert b.getWorldPtr().traversal == 1
    assert b.set_traversal(2).getWorldPtr().traversal == 2


def test_wide4_traversal_finds_the_same_hits():
    """the 4-wide rule over the same tree: the same closest hits (rounding near-ties aside); it skips the intermediate children's boxes and tests
    all four grandchildren instead, so it does MORE box tests per sample than the near-first binary walk, in about half the visits"""
    s = O.Scene.book1_final(1984)
    cam = O.camera_defocus((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, 1.5, 0.1, 10.0)
    W, H, spp = 300, 200, 4
    s.world.traversal = 0
    ref, c0 = O.render(s.world, cam, W, H, spp, 50)
    s.world.traversal = 2
    img, c2 = O.render(s.world, cam, W, H, spp, 50)
    assert np.nanmax(np.abs(img - ref)) < 1e-3
    assert c2.max_stack <= 32
    assert 0.8 * c0.box_tests < c2.box_tests < 2.0 * c0.box_tests
    print(f"box tests per sample {c0.box_tests / c0.samples:.2f} -> {c2.box_tests / c2.samples:.2f}, "
          f"leaf tests {c0.leaf_tests / c0.samples:.3f} -> {c2.leaf_tests / c2.samples:.3f}, stack {c0.max_stack} -> {c2.max_stack}")


def _comb_world(cls, node_dt, prim_dt, mat_dt, levels=12):
    """a tree that fills the 4-wide walk's stack: every level is N -> (A -> (N', leaf), B -> (leaf, leaf)), all boxes on the ray, N' nearest: a
    visit of N pushes three leaves under N', so after k levels the stack holds 3k entries; 32 are exceeded at level 11.  The binary walk
    needs 2 * levels + 2 <= 32 entries and is fine."""
    n_nodes = 6 * levels + 1
    n_leaves = 3 * levels + 1
    nodes = np.zeros(n_nodes, dtype=node_dt)
    prims = np.zeros(n_leaves, dtype=prim_dt)
    mats = np.zeros(1, dtype=mat_dt)
    mats["albedo"] = 0.5
    leaf_count = [0]

    def leaf(i):
        k = leaf_count[0]
        leaf_count[0] += 1
        nodes[i]["min"], nodes[i]["max"] = (1.0, -1.0, -1.0), (100.0, 1.0, 1.0)
        nodes[i]["left"], nodes[i]["right"] = -1, k
        prims[k]["c0"], prims[k]["radius"], prims[k]["c1"], prims[k]["mat"] = (50.0 + k * 0.01, 0.0, 0.0), 0.5, (0, 0, 0), 0

    for lv in range(levels):
        n, a, b, l1, l2, l3 = (6 * lv + j for j in range(6))
        for i in (n, a, b):
            nodes[i]["min"], nodes[i]["max"] = (0.0, -1.0, -1.0), (100.0, 1.0, 1.0)
        nodes[n]["left"], nodes[n]["right"] = a, b
        nodes[a]["left"], nodes[a]["right"] = 6 * (lv + 1), l1
        nodes[b]["left"], nodes[b]["right"] = l2, l3
        leaf(l1), leaf(l2), leaf(l3)
    leaf(6 * levels)
    nodes[6 * levels]["min"] = (0.0, -1.0, -1.0)
    w = cls()
    w.kind, w.root, w.n_nodes, w.n_prims, w.n_materials, w.max_stack = 0, 0, n_nodes, n_leaves, 1, 2 * levels + 2
    for k in range(3):
        w.bounds_min[k], w.bounds_max[k] = float(nodes[0]["min"][k]), float(nodes[0]["max"][k])
    w.nodes, w.prims, w.materials = nodes.ctypes.data, prims.ctypes.data, mats.ctypes.data
    w.traversal = 2
    return w, (nodes, prims, mats)


def test_wide4_overflow_is_reported_by_the_oracle():
    w, keep = _comb_world(O.World, O.NODE_DT, O.PRIM_DT, O.MAT_DT)
    rays = np.float32([[-10, 0, 0, 1, 0, 0, 0]])
    rc, *_ = _trace_oracle(w, rays)
    assert rc == 4
    w.traversal = 0
    rc, hit, t, prim, _ = _trace_oracle(w, rays)
    assert rc == 0 and hit[0] == 1 and prim[0] == 0
    w9, keep9 = _comb_world(O.World, O.NODE_DT, O.PRIM_DT, O.MAT_DT, levels=9)    # 3 * 9 + 1 = 28 entries: fits
    rc, hit, t, prim, _ = _trace_oracle(w9, rays)
    assert rc == 0 and hit[0] == 1 and prim[0] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("which", ["book1_final", "book2_moving", "cornell_box", "book2_final"])
def test_queue_traversal_on_the_gpu_matches_the_oracle_bit_for_bit(which, mode):
    p = pkg()
    s = config_scene(p, which).set_traversal(mode)
    w = s.getWorldPtr()
    rng = np.random.default_rng(8)
    n = 8192
    rays = random_rays(rng, n, spread=12.0 if which.startswith("book1") or which == "book2_moving" else 300.0)
    if which in ("cornell_box", "book2_final"):
        rays[:, 0:3] = np.float32([278, 278, -700]) + rng.standard_normal((n, 3)).astype(np.float32) * 40
        rays[:, 3:6] = (rng.random((n, 3), dtype=np.float32) * np.float32([555, 555, 555]) + np.float32([0, 0, 0])) - rays[:, 0:3]
    hit, t, prim, nrm = p.api.probe_trace(w, rays)
    rc, ehit, et, eprim, en = _trace_oracle(as_oracle_world(w), rays)
    assert rc == 0
    assert np.array_equal(hit, ehit) and np.array_equal(prim, eprim) and bits_equal(t, et) and bits_equal(nrm, en)
    assert hit.mean() > 0.05   # (every ray into the closed Cornell box hits something)
    # one whole sample path per key: the queue inside sample_world
    W, H = 240, 160
    cam = config_cameras(p, which, W, H)
    keys = np.stack([rng.integers(0, W * H, 1024), rng.integers(0, 16, 1024)], axis=1).astype(np.uint32)
    cfg = p.capi.RenderConfig(W, H, 16, 50, 1984, 0, 0, 1, 0)
    rad = p.api.probe_radiance(cfg, cam, w, keys)
    erad = np.zeros_like(rad)
    assert O.lib().orc_radiance_batch(C.byref(as_oracle_world(w)), C.byref(as_oracle_camera(cam)), W, H, 50, 1984, len(keys), keys, erad) == 0
    assert bits_equal(rad, erad), mismatch_report(rad, erad)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("which,W,H,spp", [("book1_final", 120, 80, 8), ("book2_moving", 96, 64, 9), ("cornell_box", 72, 72, 12), ("book2_final", 64, 40, 6)])
def test_queue_traversal_renders_on_the_streaming_kernel_bit_for_bit(which, W, H, spp, mode):
    """round 3: a RT_TRAVERSAL_QUEUE world renders on the streaming kernel (its RT_WORLD_BVH_QUEUE mode: every lane walks its trace with the
    distance-sorted queue when the trace begins, the samples are resolved in order): the framebuffer IS the oracle's; the baseline kernel
    (variant 1) still takes it, up to the summation order; the stack-walking variants refuse it"""
    p = pkg()
    s = config_scene(p, which).set_traversal(mode)
    cam = config_cameras(p, which, W, H)
    w = s.getWorldPtr()
    r = p.Renderer.MakeRenderer(W, H, spp, 50, cam, w)
    info = r.kernel_info()
    assert info["variant"] == 2 and not info["lds_resident"]
    r.Render()
    img = r.DownloadRenderbuffer()
    r.close()
    ref, _ = O.render(as_oracle_world(w), as_oracle_camera(cam), W, H, spp, 50)
    assert bits_equal(img, ref), mismatch_report(img, ref)
    r = p.Renderer.MakeRenderer(W, H, spp, 50, cam, w, variant=1)
    r.Render()
    base = r.DownloadRenderbuffer()
    r.close()
    assert np.nanmax(np.abs(base - ref)) < 1e-5 * max(1.0, float(np.nanmax(ref)))      # the baseline kernel sums a pixel's samples in another order
    for variant in (3, 4, 5):
        with pytest.raises(p.capi.RtError):
            p.Renderer.MakeRenderer(W, H, spp, 50, cam, w, variant=variant)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 2])
def test_queue_traversal_streaming_multi_pass_and_sharded(monkeypatch, mode):
    import torch
    p = pkg()
    W, H, spp = 203, 117, 10
    s = config_scene(p, "book1_final").set_traversal(mode)
    cam = config_cameras(p, "book1_final", W, H)
    w = s.getWorldPtr()
    ref, _ = O.render(as_oracle_world(w), as_oracle_camera(cam), W, H, spp, 50)
    monkeypatch.setenv("RT06_PASS_SPP", "4")
    r = p.Renderer.MakeRenderer(W, H, spp, 50, cam, w)
    r.Render()
    assert bits_equal(r.DownloadRenderbuffer(), ref)
    r.close()
    shards, last = [], None
    for rank in range(2):
        r = p.Renderer.MakeRenderer(W, H, spp, 50, cam, w, rank=rank, world_size=2)
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
    assert bits_equal(image.cpu().numpy().reshape(H, W, 4), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 2])
def test_queue_overflow_is_reported_by_the_gpu_never_silent(mode):
    p = pkg()
    make = _onion_world if mode == 1 else _comb_world
    w, keep = make(p.capi.WorldFlat, p.capi.NODE_DT, p.capi.PRIM_DT, p.capi.MAT_DT)
    rays = np.float32([[-10, 0, 0, 1, 0, 0, 0]] * 64)
    with pytest.raises(p.capi.RtError) as e:
        p.api.probe_trace(w, rays)
    assert e.value.code == 4 and ("queue" if mode == 1 else "4-wide") in str(e.value)
    # the same through the renderer (streaming kernel, queue mode): Render() reports the overflow
    cam = p.PinholeCamera((-10, 0, 0), (0, 0, 0), (0, 1, 0), 20.0, 1.0)
    r = p.Renderer.MakeRenderer(32, 32, 2, 8, cam, w)
    assert r.kernel_info()["variant"] == 2
    with pytest.raises(p.capi.RtError) as e:
        r.Render()
    assert e.value.code == 4
    r.close()
    w.traversal = 0
    hit, t, prim, _ = p.api.probe_trace(w, rays)
    assert hit.all() and (prim == 0).all()

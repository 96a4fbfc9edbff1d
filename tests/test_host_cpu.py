"""CPU tests (-m "not gpu"): the product's HOST side against the oracle, and the C-ABI surface.

No kernel is launched here: scene generation, BVH builders, flattening and camera construction run on
the host inside librt06.so and must agree bit-for-bit with the oracle's independent C restatement.
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

import _oracle as O
from _common import ROOT, as_oracle_world, bits_equal, config_cameras, oracle_scene, pkg, random_mixed_scene


def test_library_loads_and_exports_every_declared_symbol():
    p = pkg()
    L = p.lib()
    header = open(os.path.join(ROOT, "include", "rt06.h")).read()
    declared = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", header))
    declared -= {"rt_bvh_node", "rt_prim", "rt_material", "rt_world_flat", "rt_camera"}
    assert declared == set(p.capi.SYMBOLS), declared ^ set(p.capi.SYMBOLS)
    for name in declared:
        assert hasattr(L, name), f"librt06.so does not export {name}"
    assert b"gfx950" in L.rt_version()


def test_struct_layouts_match_header_sizes():
    p = pkg()
    assert p.capi.NODE_DT.itemsize == 32 and p.capi.PRIM_DT.itemsize == 32 and p.capi.MAT_DT.itemsize == 32
    assert C.sizeof(p.capi.Camera) == 76
    assert C.sizeof(p.capi.WorldFlat) == C.sizeof(O.World) == 128
    assert C.sizeof(p.capi.Camera) == C.sizeof(O.Camera)


@pytest.mark.parametrize("which", ["book1_final", "book2_moving", "three_spheres"])
def test_prefab_scene_matches_oracle_bit_for_bit(which):
    p = pkg()
    s = getattr(p.Scene, which)() if which == "three_spheres" else getattr(p.Scene, which)(1984)
    o = oracle_scene(which)
    w, ow = s.getWorldPtr(), o.world
    for f in ("kind", "root", "n_nodes", "n_prims", "n_materials", "max_stack"):
        assert getattr(w, f) == getattr(ow, f), f
    assert bits_equal(np.array(w.bounds_min[:]), np.array(ow.bounds_min[:]))
    assert bits_equal(np.array(w.bounds_max[:]), np.array(ow.bounds_max[:]))
    nodes, prims, mats = s.arrays()
    assert nodes.tobytes() == o.nodes.tobytes()
    assert prims.tobytes() == o.prims.tobytes()
    assert mats.tobytes() == o.materials.tobytes()
    if which != "three_spheres":
        assert w.n_prims == 488 and w.n_nodes == 975 and w.root == 974


@pytest.mark.parametrize("builder", [0, 1, 2, 3])
@pytest.mark.parametrize("seed", [1, 2])
def test_bvh_builders_match_oracle(builder, seed):
    """_build_bvh_rec1 / _build_bvh_rec2 / BuildBVH_BottomUp / HittableList on random sphere sets."""
    p = pkg()
    rng = np.random.default_rng(seed)
    n = 97 if builder != 2 else 41
    s = p.Scene()
    prims = np.zeros(n, dtype=O.PRIM_DT)
    mats = np.zeros(n, dtype=O.MAT_DT)
    for i in range(n):
        c0 = (rng.random(3, dtype=np.float32) * 20 - 10).astype(np.float32)
        moving = bool(i % 3 == 0)
        c1 = (c0 + rng.random(3, dtype=np.float32)).astype(np.float32) if moving else c0
        rad = np.float32(0.1 + rng.random() * 0.9)
        albedo = rng.random(3, dtype=np.float32)
        m = s.add_material(i % 3, albedo, float(np.float32(0.25)))
        (s.MakeMovingSphere(c0, c1, rad, m) if moving else s.MakeSphere(c0, rad, m))
        prims[i] = (c0, rad, c1, m | (0x80000000 if moving else 0))
        mats[i] = (albedo, np.float32(0.25), (0, 0, 0), i % 3)
    [s.BuildBVH_TopDown, s.BuildBVH_SAH, s.BuildBVH_BottomUp, s.MakeHittableList][builder]()
    o = O.Scene.from_arrays(prims, mats, builder)
    nodes, pprims, pmats = s.arrays()
    w, ow = s.getWorldPtr(), o.world
    assert (w.kind, w.root, w.n_nodes, w.max_stack) == (ow.kind, ow.root, ow.n_nodes, ow.max_stack)
    assert nodes.tobytes() == o.nodes.tobytes()
    assert pprims.tobytes() == o.prims.tobytes()
    assert pmats.tobytes() == o.materials.tobytes()
    assert bits_equal(np.array(w.bounds_min[:]), np.array(ow.bounds_min[:]))


def test_cornell_box_prefab_matches_oracle_bit_for_bit():
    """BASELINE configs[3] (extension, not in the reference): 18 quads, 4 materials, black background."""
    p = pkg()
    s, o = p.Scene.cornell_box(), O.Scene.cornell_box()
    w, ow = s.getWorldPtr(), o.world
    for f in ("kind", "root", "n_nodes", "n_prims", "n_materials", "max_stack", "n_quads", "background"):
        assert getattr(w, f) == getattr(ow, f), f
    assert (w.n_quads, w.n_prims, w.n_nodes, w.background) == (18, 0, 35, 1)
    assert bits_equal(np.array(w.background_color[:]), np.array(ow.background_color[:]))
    nodes, prims, mats = s.arrays()
    assert nodes.tobytes() == o.nodes.tobytes() and mats.tobytes() == o.materials.tobytes()
    assert s.quads().tobytes() == o.quads.tobytes()
    assert [int(m["type"]) for m in mats] == [0, 0, 0, 4]


@pytest.mark.parametrize("builder", [0, 1, 2, 3])
def test_builders_with_quads_match_oracle(builder):
    """spheres + quads through every world builder: same tree, same per-kind primitive order, same cached
    plane quantities as the oracle's restatement of quad::quad / set_bounding_box."""
    p = pkg()
    rng = np.random.default_rng(40 + builder)
    s, o = random_mixed_scene(p, rng, 23, 17, builder, background=(0.1, 0.2, 0.3))
    nodes, prims, mats = s.arrays()
    w, ow = s.getWorldPtr(), o.world
    assert (w.kind, w.root, w.n_nodes, w.max_stack, w.n_prims, w.n_quads, w.background) == \
           (ow.kind, ow.root, ow.n_nodes, ow.max_stack, ow.n_prims, ow.n_quads, ow.background)
    assert nodes.tobytes() == o.nodes.tobytes()
    assert prims.tobytes() == o.prims.tobytes()
    assert s.quads().tobytes() == o.quads.tobytes()
    assert mats.tobytes() == o.materials.tobytes()
    assert bits_equal(np.array(w.bounds_min[:]), np.array(ow.bounds_min[:]))
    assert bits_equal(np.array(w.bounds_max[:]), np.array(ow.bounds_max[:]))


def test_quads_are_refused_where_the_world_cannot_hold_them():
    p = pkg()
    s = p.Scene()
    m = s.Lambertian((0.5, 0.5, 0.5))
    with pytest.raises(p.capi.RtError):
        s.MakeQuad((0, 0, 0), (1, 0, 0), (0, 1, 0), 7)        # unknown material
    a = s.MakeSphere((0, 0, 0), 1.0, m)
    b = s.MakeSphere((3, 0, 0), 1.0, m)
    s.MakeQuad((0, 0, 0), (1, 0, 0), (0, 1, 0), m)
    root = s.bvh_node(p.Scene.prim_ref(a), p.Scene.prim_ref(b))
    with pytest.raises(p.capi.RtError):
        s.set_world_node_tree(root)                            # bvh_node trees take spheres only


def test_bvh_structure_invariants():
    p = pkg()
    s = p.Scene.book1_final(7)
    nodes, prims, _ = s.arrays()
    w = s.getWorldPtr()
    seen = np.zeros(len(prims), dtype=int)

    def walk(i, depth):
        n = nodes[i]
        if n["left"] == -1:
            seen[n["right"]] += 1
            c, r = prims[n["right"]]["c0"], prims[n["right"]]["radius"]
            assert np.all(n["min"] <= c - r) and np.all(n["max"] >= c + r)
            return depth
        for ch in (n["left"], n["right"]):
            assert ch < i  # post-order numbering: children precede parents, root is last (BVH.cu:180-210)
            assert np.all(nodes[ch]["min"] >= n["min"]) and np.all(nodes[ch]["max"] <= n["max"])
        return max(walk(n["left"], depth + 1), walk(n["right"], depth + 1))

    depth = walk(w.root, 0)
    assert np.all(seen == 1)
    assert w.max_stack == depth + 1 <= 32


@pytest.mark.parametrize("which,W,H", [("three_spheres", 400, 225), ("book1_final", 1200, 800), ("book2_moving", 800, 800)])
def test_cameras_match_oracle(which, W, H):
    p = pkg()
    cam = config_cameras(p, which, W, H)
    if which == "three_spheres":
        oc = O.camera_pinhole((0, 0, 0), (0, 0, -1), (0, 1, 0), 90.0, W / H)
    elif which == "book1_final":
        oc = O.camera_defocus((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, W / H, 0.1, 10.0)
    else:
        oc = O.camera_motion((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, W / H, 0.0, 1.0)
    assert bytes(cam) == bytes(oc)


def test_error_convention_is_never_silent():
    p = pkg()
    s = p.Scene()
    with pytest.raises(p.capi.RtError, match="no world built"):
        s.getWorldPtr()
    with pytest.raises(p.capi.RtError, match="material index"):
        s.MakeSphere((0, 0, 0), 1.0, 5)
    with pytest.raises(p.capi.RtError, match="no primitives"):
        s.BuildBVH_TopDown()
    m = s.Lambertian((0.5, 0.5, 0.5))
    s.MakeSphere((0, 0, 0), 1.0, m)
    with pytest.raises(p.capi.RtError, match="bad child"):
        s.bvh_node(5, -1)
    with pytest.raises(p.capi.RtError, match="unknown material"):
        s.add_material(9, (1, 1, 1))


def test_bvh_stack_limit_is_checked():
    """The reference's 32-entry traversal stack is never checked (BVH.cu:17,27-35); a degenerate chain
    that would overflow it must be refused at build time."""
    p = pkg()
    s = p.Scene()
    m = s.Lambertian((0.5, 0.5, 0.5))
    prims = [s.MakeSphere((float(i), 0, 0), 0.4, m) for i in range(40)]
    ref = s.prim_ref(prims[0])
    for i in range(1, 40):
        ref = s.bvh_node(ref, s.prim_ref(prims[i]))
    with pytest.raises(p.capi.RtError, match="traversal stack"):
        s.set_world_node_tree(ref)


def test_oracle_renders_config1_three_spheres():
    """BASELINE.json configs[0]: three-spheres 400x225, 1 spp, CPU reference path (plumbing, no GPU)."""
    o = oracle_scene("three_spheres")
    cam = O.camera_pinhole((0, 0, 0), (0, 0, -1), (0, 1, 0), 90.0, 400 / 225)
    img, cnt = O.render(o.world, cam, 400, 225, 1, 50)
    assert cnt.samples == 400 * 225
    assert np.all(img[..., 3] == 1.0)
    assert np.isfinite(img).all() and img.min() >= 0.0 and img.max() <= 1.0
    # sky at the top rows (row 0 = bottom), ground (yellow-green Lambertian) at the bottom
    assert img[-1, :, 2].mean() > 0.6 and img[0, :, 2].mean() < 0.3
    # thread-count independence of the oracle itself
    img1, _ = O.render(o.world, cam, 400, 225, 1, 50, threads=1)
    assert img1.tobytes() == img.tobytes()


def test_perlin_tables_match_oracle_bit_for_bit():
    """perlin::perlin() (extension): 256 unit vectors + three permutations from the host stream — product host code vs oracle."""
    p = pkg()
    s = p.Scene()
    m = s.set_perlin(77).NoiseTexture(4.0)
    s.MakeSphere((0, 0, 0), 1.0, m)
    s.BuildBVH_TopDown()
    w = s.getWorldPtr()
    got = bytes((C.c_char * 6144).from_address(w.perlin))
    o = O.Scene.three_spheres().set_perlin(77)
    assert got == o.perlin
    t = np.frombuffer(got, dtype=np.float32, count=768).reshape(256, 3)
    assert np.allclose(np.linalg.norm(t, axis=1), 1.0, atol=1e-6)
    perm = np.frombuffer(got, dtype=np.int32, offset=3072).reshape(3, 256)
    for k in range(3):
        assert sorted(perm[k].tolist()) == list(range(256)) and perm[k].tolist() != list(range(256))
    assert not np.array_equal(perm[0], perm[1])


def test_book2_final_prefab_matches_oracle_bit_for_bit():
    """BASELINE configs[4] (extension): 2401 quads, 1008 spheres, 10 materials, Perlin tables, synthetic planet image."""
    p = pkg()
    s, o = p.Scene.book2_final(1984), O.Scene.book2_final(1984)
    w, ow = s.getWorldPtr(), o.world
    for f in ("kind", "root", "n_nodes", "n_prims", "n_materials", "max_stack", "n_quads", "background", "image_width", "image_height"):
        assert getattr(w, f) == getattr(ow, f), f
    assert (w.n_prims, w.n_quads, w.n_materials, w.n_nodes) == (1008, 2401, 10, 2 * 3409 - 1)
    nodes, prims, mats = s.arrays()
    assert nodes.tobytes() == o.nodes.tobytes() and prims.tobytes() == o.prims.tobytes() and mats.tobytes() == o.materials.tobytes()
    assert s.quads().tobytes() == o.quads.tobytes()
    assert s.perlin_bytes() == o.perlin
    img = s.image()
    oimg = bytes((C.c_char * (256 * 128 * 3)).from_address(ow.image))
    assert img.shape == (128, 256, 3) and img.tobytes() == oimg
    assert sorted(int(t) for t in mats["type"]) == [0, 0, 0, 1, 2, 4, 5, 5, 6, 7]


def test_bench_workload_table_builds_matching_product_and_oracle_scenes():
    """bench.py --workload: every entry must construct the same world through the product host code and the oracle."""
    import argparse
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    p = pkg()
    assert set(bench.WORKLOADS) == {"book1_final", "book2_moving", "cornell_box", "book2_final"}
    for name, wl in bench.WORKLOADS.items():
        args = argparse.Namespace(workload=name, width=wl[4], height=wl[5], spp=wl[6], depth=wl[7], seed=1984)
        scene, cam = bench.make_workload(args, p)
        oscene, ocam = bench.make_workload(args, O, oracle=True)
        w, ow = scene.getWorldPtr(), oscene.world
        assert (w.kind, w.n_nodes, w.n_prims, w.n_quads, w.n_materials) == (ow.kind, ow.n_nodes, ow.n_prims, ow.n_quads, ow.n_materials), name
        assert scene.arrays()[0].tobytes() == oscene.nodes.tobytes(), name
        assert bytes(cam) == bytes(ocam), name


def test_bench_roofline_is_recomputable_from_the_committed_counter_summary():
    """bench.py's `roofline` = SQ_INSTS_VALU of the newest profiles/rNN_bench_pmc_summary.csv / the live kernel time against
    SIMDs x clock / 2: recompute it here from the committed files alone and require frac <= 1 and a source-hash stamp."""
    import csv
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    v, src, stamp, at_spp = None, None, None, None
    for kernel in ("render_kernel_xchg", "render_kernel_stream"):   # whichever kernel the newest headline summary profiled
        v, src, stamp, at_spp = bench.profiled_counters(kernel, "book1_final", 1200, 800, 500, 50)
        newest = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_bench_pmc_summary.csv"))[-1]
        if v is not None and src == f"profiles/{newest}":
            break
    assert v is not None and src.startswith("profiles/r") and stamp and len(stamp) == 64 and at_spp == 500
    stats = [r for r in csv.DictReader(open(os.path.join(ROOT, src.replace("_pmc_summary", "_kernel_stats")))) if kernel in r["Name"]]
    kernel_ms = float(stats[0]["AverageNs"]) / 1e6
    roof = bench.issue_roofline(v, kernel_ms, 1024, 2.4)
    assert 0.5 < roof["frac"] <= 1.0 and 0.4 < roof["lanes_active_frac"] < 1.0 and roof["lane_weighted_frac"] < roof["frac"]
    assert abs(roof["achieved"] - v["SQ_INSTS_VALU"] / (kernel_ms * 1e-3) / 1e9) < 0.01 and roof["peak"] == 1228.8
    # a summary of another workload, taken at another spp, is found by workload + frame size + depth and scaled to the asked spp
    v2, src2, _, at2 = bench.profiled_counters("render_kernel_stream", "book2_final", 800, 800, 400, 40)
    assert src2.endswith("book2_final_pmc_summary.csv") and at2 == 200
    v200, _, _, _ = bench.profiled_counters("render_kernel_stream", "book2_final", 800, 800, 200, 40)
    assert abs(v2["SQ_INSTS_VALU"] - 2.0 * v200["SQ_INSTS_VALU"]) < 1e-6 * v2["SQ_INSTS_VALU"]
    assert bench.profiled_counters("render_kernel_stream", "book2_final", 800, 801, 200, 40)[0] is None
    # the bench line committed next to it carries the same kind of object
    line = json.load(open(os.path.join(ROOT, src.replace("_pmc_summary.csv", ".json"))))
    assert line["roofline"]["bound"] == "valu_issue" and line["roofline"]["frac"] <= 1.0 and line["parity"]["bit_identical"]
    assert line["metric"].startswith("Msamples/sec") and line["scaling"] == "strong" and line["cpu_baseline"]["kind"] == "port"

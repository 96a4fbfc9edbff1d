"""Pins against the REFERENCE itself.

tests/golden/glm_*   : the reference's vendored GLM 0.9.9.7 + utilities/glm_utils.h   (oracle/ref_glm_probe.cpp)
tests/golden/ref_*   : the reference's own rt_engine headers — ray_data.cuh, geometry/aabb.cuh, HittableList.cuh,
                       bvh_node.cuh, BVH.cuh (layout), shaders/cu_Textures.cuh — compiled by plain g++ against NVIDIA's
                       real <cuda_runtime.h> from the image's triton wheel      (oracle/ref_path_probe.cpp)
Both sets are produced by `python oracle/gen_golden.py` in the dev container and committed as data.

CPU tests (`not gpu`): the oracle (and the product's host-side aabb helpers) reproduce the reference's outputs bit for bit.
GPU tests: the HIP device functions reproduce the same outputs bit for bit, called through the C ABI probes — the direct
reference -> HIP check.  What stays unpinned (the reference files do not build here: <format>, <<<>>>, cuRAND):
_sphere_closest_intersection, BVH::ClosestIntersection, Scatter, the cameras, sample_world, render_kernel.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

import _oracle as O
from _common import bits_equal, mismatch_report, pkg

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MISS = np.float32(3.402823466e+38)


def gold(name, cols, dtype="<f4"):
    return np.fromfile(os.path.join(GOLD, name), dtype=dtype).reshape(-1, cols)


def olib():
    L = O.lib()
    L.orc_aabb_misc_batch.argtypes = [C.c_size_t, O.f32p, O.f32p]
    L.orc_checker_batch.argtypes = [C.c_size_t, O.f32p, O.f32p]
    L.orc_ray_batch.argtypes = [C.c_size_t, O.f32p, O.f32p]
    L.orc_trace_counts.argtypes = [C.POINTER(O.World), C.c_size_t, O.f32p, O.u32p, O.u32p]
    return L


# ---------------------------------------------------------------------------------------------------------------------
# the aggregate fixtures as flat worlds (64 scenarios x 8 spheres x 64 rays)
# ---------------------------------------------------------------------------------------------------------------------
class AggScenario:
    """One scenario of tests/golden/ref_agg_*: arrays kept alive + a world struct of the requested binding class."""

    def __init__(self, s, kind, world_cls, node_dt, prim_dt, mat_dt):
        spheres = gold("ref_agg_spheres.f32", 4).reshape(64, 8, 4)[s]
        self.prims = np.zeros(8, dtype=prim_dt)
        self.prims["c0"], self.prims["radius"], self.prims["c1"], self.prims["mat"] = spheres[:, :3], spheres[:, 3], spheres[:, :3], 0
        self.mats = np.zeros(1, dtype=mat_dt)
        self.mats["albedo"] = 0.5
        boxes = gold("ref_agg_nodeboxes.f32", 6).reshape(64, 7, 6)[s]
        refs = gold("ref_agg_refs.i32", 15, "<i4")[s]
        self.nodes = np.zeros(7, dtype=node_dt)
        self.nodes["min"], self.nodes["max"] = boxes[:, :3], boxes[:, 3:]
        self.nodes["left"], self.nodes["right"] = refs[0:14:2], refs[1:14:2]
        lb = gold("ref_agg_listbounds.f32", 6)[s]
        w = world_cls()
        w.kind = 1 if kind == "list" else 2   # RT_WORLD_LIST / RT_WORLD_NODE_TREE
        w.root = int(refs[14]) if kind == "tree" else 0
        w.n_nodes = 7 if kind == "tree" else 0
        w.n_prims, w.n_materials, w.max_stack = 8, 1, 8
        # a HittableList carries its own bounds (HittableList.cuh:19); a bvh_node tree's root box is node data
        for k in range(3):
            w.bounds_min[k], w.bounds_max[k] = (float(lb[k]), float(lb[3 + k])) if kind == "list" else (float(boxes[int(refs[14])][k]), float(boxes[int(refs[14])][3 + k]))
        w.nodes = self.nodes.ctypes.data if kind == "tree" else None
        w.prims, w.materials = self.prims.ctypes.data, self.mats.ctypes.data
        self.world = w
        self.rays = np.ascontiguousarray(gold("ref_agg_rays.f32", 7).reshape(64, 64, 7)[s])
        self.expect = gold(f"ref_agg_{kind}_out.f32", 12).reshape(64, 64, 12)[s]   # hit, t, prim, n_visits, order[8]


def check_agg(kind, trace_fn, world_cls, node_dt, prim_dt, mat_dt, counts_fn=None):
    n_hits = 0
    for s in range(64):
        sc = AggScenario(s, kind, world_cls, node_dt, prim_dt, mat_dt)
        hit, t, prim = trace_fn(sc.world, sc.rays)
        e = sc.expect
        assert np.array_equal(hit, e[:, 0].astype(np.int32)), f"{kind} scenario {s}: hit flags differ at rays {np.nonzero(hit != e[:, 0])[0][:5]}"
        assert bits_equal(t, e[:, 1]), f"{kind} scenario {s}: " + mismatch_report(t, e[:, 1])
        assert np.array_equal(prim, e[:, 2].astype(np.int32)), f"{kind} scenario {s}: closest primitive differs (visiting order / tie rule)"
        if counts_fn is not None:
            leaf = counts_fn(sc.world, sc.rays)
            assert np.array_equal(leaf, e[:, 3].astype(np.uint32)), f"{kind} scenario {s}: number of leaves reached differs"
        n_hits += int(hit.sum())
    assert n_hits > 1000


# ---------------------------------------------------------------------------------------------------------------------
# CPU: oracle (and the product's host helpers) against the reference
# ---------------------------------------------------------------------------------------------------------------------
def test_reference_record_layouts():
    """sizeof / offsetof of the reference's PODs, as compiled from its headers, against the flat records of include/rt06.h."""
    lay = json.load(open(os.path.join(GOLD, "ref_layout.json")))
    assert lay["sizeof_Ray"] == 28 and lay["offsetof_Ray_d"] == 12 and lay["offsetof_Ray_time"] == 24
    assert lay["sizeof_RayPayload"] == 40 and lay["sizeof_aabb"] == 24
    p = pkg()
    assert lay["sizeof_BVH_Node"] == p.capi.NODE_DT.itemsize == O.NODE_DT.itemsize == 32
    assert lay["offsetof_BVH_Node_left_child_idx"] == p.capi.NODE_DT.fields["left"][1] == 24
    assert lay["offsetof_BVH_Node_right_child_hittable_idx"] == p.capi.NODE_DT.fields["right"][1] == 28
    assert lay["IS_LEAF_CODE"] == -1
    assert np.array([lay["MISS_DIST_bits"]], np.uint32).view(np.float32)[0] == MISS


def test_oracle_aabb_intersects_matches_reference():
    """G1: aabb::intersects (aabb.cuh:30-44) incl. 0 / +-inf / NaN / origin-inside / inverted / empty boxes."""
    i, e = gold("ref_aabb_in.f32", 13), gold("ref_aabb_out.f32", 2)
    n = len(i)
    hit, dist = np.zeros(n, np.int32), np.zeros(n, np.float32)
    O.lib().orc_aabb_batch(n, np.ascontiguousarray(i[:, 0:6]), np.ascontiguousarray(i[:, 6:12]), np.ascontiguousarray(i[:, 12]), hit, dist)
    assert np.array_equal(hit, e[:, 0].astype(np.int32)), f"{(hit != e[:, 0]).sum()} hit flags differ"
    assert bits_equal(dist, e[:, 1]), mismatch_report(dist, e[:, 1])
    assert 0.2 < e[:, 0].mean() < 0.7


def test_oracle_aabb_helpers_match_reference():
    i, e = gold("ref_aabbmisc_in.f32", 12), gold("ref_aabbmisc_out.f32", 20)
    out = np.zeros_like(e)
    olib().orc_aabb_misc_batch(len(i), i, out)
    assert bits_equal(out, e), mismatch_report(out, e)


def test_product_host_aabb_helpers_match_reference():
    """the builders' helpers inside librt06.so (csrc/rt_host.cpp), through the C ABI — runs without a GPU"""
    i, e = gold("ref_aabbmisc_in.f32", 12), gold("ref_aabbmisc_out.f32", 20)
    out = pkg().api.probe_aabb_misc(i)
    assert bits_equal(out, e), mismatch_report(out, e)


def test_oracle_checker_texture_matches_reference():
    """checker_texture::value (cu_Textures.cuh:31-39): truncation toward zero, negative coordinates, cell boundaries"""
    i, e = gold("ref_checker_in.f32", 10), gold("ref_checker_out.f32", 3)
    out = np.zeros_like(e)
    olib().orc_checker_batch(len(i), i, out)
    assert bits_equal(out, e), mismatch_report(out, e)
    assert 0.3 < np.all(e == i[:, 0:3], axis=1).mean() < 0.7   # both colours occur


def test_oracle_ray_at_and_backfacing_match_reference():
    i, e = gold("ref_ray_in.f32", 10), gold("ref_ray_out.f32", 4)
    out = np.zeros_like(e)
    olib().orc_ray_batch(len(i), i, out)
    assert bits_equal(out, e), mismatch_report(out, e)


def _oracle_trace(world, rays):
    n = len(rays)
    hit, t, prim, nrm = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros(n, np.int32), np.zeros((n, 3), np.float32)
    assert O.lib().orc_trace_batch(C.byref(world), n, rays, hit, t, prim, nrm) == 0
    return hit, t, prim


def _oracle_counts(world, rays):
    leaf, box = np.zeros(len(rays), np.uint32), np.zeros(len(rays), np.uint32)
    assert olib().orc_trace_counts(C.byref(world), len(rays), rays, leaf, box) == 0
    return leaf


@pytest.mark.parametrize("kind", ["list", "tree"])
def test_oracle_aggregates_match_reference(kind):
    """HittableList::ClosestIntersection (HittableList.cuh:21-34) / bvh_node::ClosestIntersection (bvh_node.cuh:19-24):
    closest hit, its primitive (identical spheres expose the visiting order) and the number of leaves reached."""
    check_agg(kind, _oracle_trace, O.World, O.NODE_DT, O.PRIM_DT, O.MAT_DT, _oracle_counts)


def test_reference_fixture_visit_orders_are_consistent():
    """the recorded leaf order of the list is 0..7 whenever the bounds pre-test passes; of the tree, a pre-order walk"""
    e = gold("ref_agg_list_out.f32", 12)
    visited = e[:, 3] > 0
    assert np.all(e[visited, 3] == 8) and np.all(e[visited, 4:12] == np.arange(8, dtype=np.float32))
    assert 0.05 < (~visited).mean() < 0.6


# ---------------------------------------------------------------------------------------------------------------------
# GPU: the HIP device functions against the reference
# ---------------------------------------------------------------------------------------------------------------------
GLM_SHAPES = {"dot": (6, 1), "cross": (6, 3), "normalize": (3, 3), "reflect": (6, 3), "refract": (7, 3), "mix3": (7, 3), "mix1": (3, 1),
              "min3": (6, 3), "max3": (6, 3), "compmax": (3, 1), "compmin": (3, 1), "clamp01_sqrt": (3, 3), "near_zero": (3, 1),
              "length2": (3, 1), "lerp": (7, 3), "radians": (1, 1)}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(GLM_SHAPES))
def test_device_math_matches_reference_glm(name):
    """csrc/rt_math.hpp on gfx950 == the reference's GLM / glm_utils.h on the committed vectors (NaN / inf / denormals included)"""
    nin, nout = GLM_SHAPES[name]
    i, e = gold(f"glm_{name}_in.f32", nin), gold(f"glm_{name}_out.f32", nout)
    assert len(i) == len(e) == 1024
    out = pkg().api.probe_glm(name, i)
    assert bits_equal(out, e), f"{name}: " + mismatch_report(out, e)


@pytest.mark.gpu
def test_device_ray_at_and_backfacing_match_reference():
    i, e = gold("ref_ray_in.f32", 10), gold("ref_ray_out.f32", 4)
    out = pkg().api.probe_glm("ray", i)
    assert bits_equal(out, e), mismatch_report(out, e)


@pytest.mark.gpu
def test_device_aabb_intersects_matches_reference():
    i, e = gold("ref_aabb_in.f32", 13), gold("ref_aabb_out.f32", 2)
    hit, dist = pkg().api.probe_aabb(i[:, 0:6], i[:, 6:12], i[:, 12])
    assert np.array_equal(hit, e[:, 0].astype(np.int32)), f"{(hit != e[:, 0]).sum()} hit flags differ"
    assert bits_equal(dist, e[:, 1]), mismatch_report(dist, e[:, 1])


@pytest.mark.gpu
def test_device_checker_texture_matches_reference():
    """LambertianTexture's attenuation = checker_texture::value at the hit point (cu_materials.cuh:27-40): the scatter probe with
    the hit point placed at `pos` (origin = pos, distance 0)"""
    p = pkg()
    i, e = gold("ref_checker_in.f32", 10), gold("ref_checker_out.f32", 3)
    n = len(i)
    mats = np.zeros(n, dtype=p.capi.MAT_DT)
    mats["albedo"], mats["albedo2"], mats["type"] = i[:, 0:3], i[:, 3:6], p.capi.MAT_LAMBERTIAN_CHECKER
    mats["param"] = np.float32(1.0) / i[:, 6]                # inv_scale(1.0f / scale), cu_Textures.cuh:27
    rays = np.zeros((n, 7), np.float32)
    rays[:, 0:3], rays[:, 3:6] = i[:, 7:10], (0.0, -1.0, 0.0)
    normals = np.tile(np.float32([0, 1, 0]), (n, 1))
    keys = np.stack([np.arange(n, dtype=np.uint32), np.zeros(n, np.uint32)], axis=1)
    sc, _, att, _ = p.api.probe_scatter(7, mats, rays, np.zeros(n, np.float32), normals, keys)
    ok = sc == 1                                           # a degenerate direction absorbs (cu_materials.cuh:34): no colour then
    assert ok.mean() > 0.99
    assert bits_equal(att[ok], e[ok]), mismatch_report(att[ok], e[ok])


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["list", "tree"])
def test_device_aggregates_match_reference(kind):
    p = pkg()

    def trace(world, rays):
        hit, t, prim, _ = p.api.probe_trace(world, rays)
        return hit, t, prim
    check_agg(kind, trace, p.capi.WorldFlat, p.capi.NODE_DT, p.capi.PRIM_DT, p.capi.MAT_DT)

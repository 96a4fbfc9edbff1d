"""Shared helpers for the tests: load the product package and build matching oracle/product inputs."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as G  # noqa: E402
import _oracle as O  # noqa: E402


def pkg():
    return G.load_package()


def bits_equal(a, b):
    """bit-exact comparison of float arrays; NaNs compare equal to NaNs."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    if a.shape != b.shape:
        return False
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


def mismatch_report(a, b, names=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    bad = ~((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b)))
    idx = np.argwhere(bad)
    return f"{bad.sum()} of {bad.size} differ; first at {idx[:3].tolist()}: got {a[bad][:3]} expected {b[bad][:3]}"


def as_oracle_world(world_flat):
    """Reinterpret a product rt_world_flat as an oracle orc_world (identical byte layout by contract)."""
    assert C.sizeof(world_flat) == C.sizeof(O.World)
    w = O.World()
    C.memmove(C.byref(w), C.byref(world_flat), C.sizeof(O.World))
    return w


def as_oracle_camera(cam):
    assert C.sizeof(cam) == C.sizeof(O.Camera)
    c = O.Camera()
    C.memmove(C.byref(c), C.byref(cam), C.sizeof(O.Camera))
    return c


# the benchmark configurations of BASELINE.json (scene, camera, W, H, spp); depth 50 everywhere
def config_cameras(p, which, W, H):
    if which == "three_spheres":  # config 1
        return p.PinholeCamera((0, 0, 0), (0, 0, -1), (0, 1, 0), 90.0, W / H)
    if which == "book1_final":    # config 2
        return p.DefocusBlurCamera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, W / H, 0.1, 10.0)
    if which == "book2_moving":   # config 3
        return p.MotionBlurCamera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, W / H, 0.0, 1.0)
    if which == "cornell_box":    # config 4 (not in the reference): the book's Cornell camera
        return p.PinholeCamera((278, 278, -800), (278, 278, 0), (0, 1, 0), 40.0, W / H)
    if which == "book2_final":    # config 5 (not in the reference): the book's final-scene camera, shutter 0..1
        return p.MotionBlurCamera((478, 278, -600), (278, 278, 0), (0, 1, 0), 40.0, W / H, 0.0, 1.0)
    raise ValueError(which)


def config_scene(p, which, seed=1984):
    return getattr(p.Scene, which)() if which in ("three_spheres", "cornell_box") else getattr(p.Scene, which)(seed)


def random_mixed_scene(p, rng, n_spheres, n_quads, builder, background=None, lights=True, media=False):
    """The same random spheres + quads (+ lights, + constant media) through the product vocabulary and the oracle's arrays."""
    s = p.Scene()
    n_mats = 8 if media else 6
    mats = np.zeros(n_mats, dtype=O.MAT_DT)
    for i in range(n_mats):
        albedo = rng.random(3, dtype=np.float32)
        mtype = ([0, 1, 2, 0, 1, 4, 5, 5][i] if lights else [0, 1, 2, 0, 1, 0, 5, 5][i])
        if mtype == 4:
            albedo = (albedo * np.float32(6.0)).astype(np.float32)
        param = np.float32([0.0, 0.3, 1.5, 0.0, 0.0, 0.0, 0.8, 0.05][i])
        s.add_material(mtype, albedo, float(param))
        mats[i] = (albedo, param, (0, 0, 0), mtype)
    prims = np.zeros(n_spheres, dtype=O.PRIM_DT)
    for i in range(n_spheres):
        c0 = (rng.random(3, dtype=np.float32) * 10 - 5).astype(np.float32)
        moving = bool(i % 4 == 0)
        c1 = (c0 + rng.random(3, dtype=np.float32) * np.float32(0.5)).astype(np.float32) if moving else c0
        rad = np.float32(0.2 + rng.random() * 0.8)
        m = int(rng.integers(0, n_mats))
        if m == 7:
            rad = np.float32(rad * 6)   # the thin medium is a large ball of haze
        (s.MakeMovingSphere(c0, c1, rad, m) if moving else s.MakeSphere(c0, rad, m))
        prims[i] = (c0, rad, c1, m | (0x80000000 if moving else 0))
    quads = np.zeros(n_quads, dtype=O.QUAD_DT)
    for i in range(n_quads):
        Q = (rng.random(3, dtype=np.float32) * 12 - 6).astype(np.float32)
        if i % 3 == 0:  # axis-aligned like the Cornell walls (zero-thickness boxes get padded)
            u = np.float32([rng.random() * 4 + 0.5, 0, 0]); v = np.float32([0, 0, rng.random() * 4 + 0.5])
        else:
            u = (rng.standard_normal(3) * 2).astype(np.float32); v = (rng.standard_normal(3) * 2).astype(np.float32)
        m = int(rng.integers(0, 6))   # a quad never bounds a medium
        s.MakeQuad(Q, u, v, m)
        quads[i]["Q"], quads[i]["u"], quads[i]["v"], quads[i]["mat"] = Q, u, v, m
    if background is not None:
        s.set_background(background)
    [s.BuildBVH_TopDown, s.BuildBVH_SAH, s.BuildBVH_BottomUp, s.MakeHittableList][builder]()
    o = O.Scene.from_arrays_ext(prims, quads, mats, builder, 0 if background is None else 1,
                                (0, 0, 0) if background is None else background)
    return s, o


def oracle_scene(which, seed=1984):
    return getattr(O.Scene, which)() if which == "three_spheres" else getattr(O.Scene, which)(seed)


def random_rays(rng, n, with_time=True, spread=15.0):
    o = (rng.random((n, 3), dtype=np.float32) * 2 - 1) * spread
    o[:, 1] = np.abs(o[:, 1]) * 0.3 + 0.05
    d = rng.standard_normal((n, 3)).astype(np.float32)
    cols = [o, d]
    if with_time:
        cols.append(rng.random((n, 1), dtype=np.float32))
    return np.ascontiguousarray(np.concatenate(cols, axis=1), dtype=np.float32)

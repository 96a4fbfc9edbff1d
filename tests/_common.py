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
    raise ValueError(which)


def config_scene(p, which, seed=1984):
    return getattr(p.Scene, which)() if which == "three_spheres" else getattr(p.Scene, which)(seed)


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

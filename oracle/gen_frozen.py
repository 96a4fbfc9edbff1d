#!/usr/bin/env python3
"""Freeze outputs of the CPU oracle as committed fixtures tests/golden/frozen_*.npz (test infrastructure).

The `-m gpu` parity tests compare the HIP path with the oracle LIVE; if oracle and kernels ever drifted together nothing
would notice.  These files pin today's oracle outputs on seeded inputs: the sphere test, the three
cameras, Scatter, BVH::ClosestIntersection on the two Book scenes (with the scenes' own bytes: host RNG stream, builder),
per-sample radiance for fixed (pixel, sample) keys, and the config-1 image.  tests/test_frozen_fixtures.py checks the
oracle (CPU) and the HIP probes / renderer (GPU) against them bit for bit.

They are ORACLE outputs, not reference outputs: what of the path is pinned to the reference itself is in glm_* / ref_*
(oracle/gen_golden.py).  Regenerate with `python oracle/gen_frozen.py` — byte-identical unless the oracle changed, and a
change must be explained in the commit that updates them.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import _oracle as O  # noqa: E402

OUT = os.path.join(HERE, "..", "tests", "golden")
L = O.lib()
SEED = 1984


def save(name, **arrays):
    path = os.path.join(OUT, f"frozen_{name}.npz")
    np.savez_compressed(path, **arrays)
    print(f"{path}: {os.path.getsize(path) / 1024:.0f} KB  [{', '.join(arrays)}]")


def cam_bytes(c):
    return np.frombuffer(bytes(c), dtype=np.uint8).copy()


def rays6(rng, n, spread):
    o = ((rng.random((n, 3), dtype=np.float32) * 2 - 1) * spread).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    return np.ascontiguousarray(np.concatenate([o, d], axis=1))


# ---- G2: _sphere_closest_intersection: tangent, origin inside, behind, un-normalised directions ------------------------
rng = np.random.default_rng(2)
n = 4096
sph = np.concatenate([(rng.random((n, 3), dtype=np.float32) * 4 - 2), (rng.random((n, 1), dtype=np.float32) * 1.5 + 0.05)], axis=1).astype(np.float32)
r6 = rays6(rng, n, 4.0)
aim = sph[:, :3] + rng.standard_normal((n, 3)).astype(np.float32) * sph[:, 3:4] * np.float32(0.7)
r6[:, 3:6] = aim - r6[:, 0:3]
r6[::8, 0:3] = sph[::8, :3] + (rng.random((n // 8, 3), dtype=np.float32) - 0.5) * sph[::8, 3:4]       # origin inside
r6[1::8, 3:6] *= -1                                                                                    # sphere behind
r6[2::8, 3:6] *= np.ldexp(np.float32(1), rng.integers(-12, 12, (n // 8, 1))).astype(np.float32)      # any length
t = np.zeros(n, np.float32)
L.orc_sphere_batch(n, r6, sph, t)
save("sphere", rays=r6, spheres=sph, t=t)

# ---- G3: sample_ray of the three cameras (+ the camera PODs the constructors produce) ----------------------------------
rng = np.random.default_rng(3)
n = 2048
st = (rng.random((n, 2), dtype=np.float32) * 2 - 1).astype(np.float32)
keys = np.stack([rng.integers(0, 960000, n), rng.integers(0, 500, n)], axis=1).astype(np.uint32)
cams = {"pinhole": O.camera_pinhole((0, 0, 0), (0, 0, -1), (0, 1, 0), 90.0, 400 / 225),
        "defocus": O.camera_defocus((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, 1.5, 0.1, 10.0),
        "motion": O.camera_motion((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, 1.0, 0.0, 1.0)}
arrs = dict(st=st, keys=keys)
for name, c in cams.items():
    orays, draws = np.zeros((n, 7), np.float32), np.zeros(n, np.uint32)
    L.orc_camera_batch(SEED, C.byref(c), n, st, keys, orays, draws)
    arrs[f"{name}_pod"], arrs[f"{name}_rays"], arrs[f"{name}_draws"] = cam_bytes(c), orays, draws
save("cameras", **arrs)

# ---- G4: Scatter x {Lambertian, Metal fuzz 0 / 0.3 / 1, Dielectric front / back / TIR, checker} -------------------------
rng = np.random.default_rng(4)
n = 4096
mats = np.zeros(n, dtype=O.MAT_DT)
mats["albedo"] = rng.random((n, 3), dtype=np.float32)
mats["albedo2"] = rng.random((n, 3), dtype=np.float32)
mats["type"] = np.arange(n) % 4
metal = mats["type"] == 1
mats["param"][metal] = rng.choice(np.array([0.0, 0.3, 1.0], np.float32), metal.sum())
diel = mats["type"] == 2
mats["param"][diel] = rng.choice(np.array([1.5, 1.0 / 1.5, 1.333, 2.4], np.float32), diel.sum())
mats["param"][mats["type"] == 3] = np.float32(1.0) / np.float32(0.32)
normals = rng.standard_normal((n, 3)).astype(np.float32)
normals /= np.linalg.norm(normals, axis=1, keepdims=True).astype(np.float32)
rays = np.concatenate([rays6(rng, n, 5.0), rng.random((n, 1), dtype=np.float32)], axis=1).astype(np.float32)
rays[1::2, 3:6] = (-normals[1::2] + rng.standard_normal((n // 2, 3)).astype(np.float32) * np.float32(0.8))   # mostly front-facing
rays[3::16, 3:6] = normals[3::16] * np.float32(0.05) + np.cross(normals[3::16], np.float32([0.3, 0.5, 0.7]))  # grazing from inside: TIR
dist = (rng.random(n, dtype=np.float32) * 10).astype(np.float32)
keys = np.stack([rng.integers(0, 960000, n), rng.integers(0, 500, n)], axis=1).astype(np.uint32)
sc, orays, att, draws = np.zeros(n, np.int32), np.zeros((n, 7), np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.uint32)
L.orc_scatter_batch(SEED, n, mats.ctypes.data, np.ascontiguousarray(rays), dist, np.ascontiguousarray(normals), keys, sc, orays, att, draws)
save("scatter", mats=mats.view(np.uint8).reshape(n, -1), rays=rays, dist=dist, normals=normals, keys=keys, scattered=sc, out_rays=orays, atten=att, draws=draws)

# ---- G5 + G6: the two Book scenes: bytes of the scene, closest hits of 4096 rays, radiance of 1024 (pixel, sample) keys ---
for which, cam in (("book1_final", cams["defocus"]), ("book2_moving", O.camera_motion((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, 1.0, 0.0, 1.0))):
    scene = getattr(O.Scene, which)(SEED)
    rng = np.random.default_rng(5)
    n = 4096
    rays = np.concatenate([rays6(rng, n, 12.0), rng.random((n, 1), dtype=np.float32)], axis=1).astype(np.float32)
    rays[:, 1] = np.abs(rays[:, 1]) * np.float32(0.3) + np.float32(0.05)
    rays[: n // 2, 4] = -np.abs(rays[: n // 2, 4])
    rays[:64, 3] = 0.0
    rays = np.ascontiguousarray(rays)
    hit, t, prim, nrm = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros(n, np.int32), np.zeros((n, 3), np.float32)
    assert L.orc_trace_batch(C.byref(scene.world), n, rays, hit, t, prim, nrm) == 0
    W, H = (1200, 800) if which == "book1_final" else (800, 800)
    nk = 1024
    keys = np.stack([rng.integers(0, W * H, nk), rng.integers(0, 500, nk)], axis=1).astype(np.uint32)
    rad = np.zeros((nk, 3), np.float32)
    assert L.orc_radiance_batch(C.byref(scene.world), C.byref(cam), W, H, 50, SEED, nk, keys, rad) == 0
    save(which, nodes=scene.nodes.view(np.uint8).reshape(len(scene.nodes), -1), prims=scene.prims.view(np.uint8).reshape(len(scene.prims), -1),
         materials=scene.materials.view(np.uint8).reshape(len(scene.materials), -1), root=np.int32(scene.world.root),
         rays=rays, hit=hit, t=t, prim=prim, normal=nrm, camera_pod=cam_bytes(cam), width=np.int32(W), height=np.int32(H), keys=keys, radiance=rad)

# ---- G7: the config-1 image: three spheres, 400 x 225, 1 spp, depth 50 (BASELINE.json configs[0]) ------------------------
scene = O.Scene.three_spheres()
img, _ = O.render(scene.world, cams["pinhole"], 400, 225, 1, 50, SEED)
save("config1_image", rgb=np.ascontiguousarray(img[..., :3]), alpha_all_one=np.bool_(np.all(img[..., 3] == 1.0)))

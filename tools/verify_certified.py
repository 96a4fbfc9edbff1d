#!/usr/bin/env python3
"""The certified far planes of the default hot loop against aabb::intersects on many adversarial box pairs (rt_probe_boxpair_certified; the generator of
tests/test_gpu_parity.py::test_certified_far_planes_take_the_exact_decisions, other seeds): rays aimed exactly at corners and edge points, flat boxes,
origins on a face, sibling boxes that share faces, rec_t on the entry distance.  Prints the number of cases, of exact redos and of differing decisions.
    python tools/verify_certified.py [chunks of 2^21 cases, default 32]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

p = G.load_package()
chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = 1 << 21
tot = reg_tot = redo = bad = 0
for c in range(chunks):
    rng = np.random.default_rng(1000 + c)
    scale = np.float32(rng.choice([0.01, 1.0, 12.0, 500.0, 1e5]))
    lo = ((rng.random((n, 3), dtype=np.float32) * 2 - 1) * scale).astype(np.float32)
    ext = (rng.random((n, 3), dtype=np.float32) * np.float32(0.25) * scale + np.float32(1e-3) * scale).astype(np.float32)
    ext[rng.random((n, 3)) < 0.03] = 0.0
    left = np.concatenate([lo, lo + ext], axis=1)
    shift = np.where(rng.random((n, 3)) < 0.5, 0.0, rng.random((n, 3)) * 2 - 1).astype(np.float32)
    rlo = (lo + shift * ext).astype(np.float32)
    rext = np.where(rng.random((n, 3)) < 0.5, ext, (rng.random((n, 3), dtype=np.float32) * np.float32(0.25) * scale + np.float32(1e-3) * scale)).astype(np.float32)
    right = np.concatenate([rlo, rlo + rext], axis=1)
    boxes = np.ascontiguousarray(np.concatenate([left, right], axis=1), dtype=np.float32)
    rays = np.empty((n, 6), np.float32)
    rays[:, 0:3] = (rng.random((n, 3), dtype=np.float32) * 2 - 1) * scale * 2
    rays[:, 3:6] = rng.standard_normal((n, 3)).astype(np.float32) * np.float32(rng.choice([1e-3, 1.0, 50.0]))
    q = n // 4
    corner = np.where(rng.random((q, 3)) < 0.5, left[:q, 0:3], left[:q, 3:6]).astype(np.float32)
    rays[:q, 3:6] = (corner - rays[:q, 0:3]) * (rng.random((q, 1), dtype=np.float32) + 0.5)
    edge = np.where(rng.random((q, 3)) < 0.5, right[q:2 * q, 0:3], right[q:2 * q, 3:6]).astype(np.float32)
    free = rng.integers(0, 3, q)
    t = rng.random(q, dtype=np.float32)
    ar = np.arange(q)
    edge[ar, free] = (right[q:2 * q, 0:3][ar, free] * (1 - t) + right[q:2 * q, 3:6][ar, free] * t).astype(np.float32)
    rays[q:2 * q, 3:6] = (edge - rays[q:2 * q, 0:3]) * (rng.random((q, 1), dtype=np.float32) * 3 + 0.25)
    k = np.arange(2 * q, 2 * q + n // 8)
    ax = rng.integers(0, 3, len(k))
    rays[k, ax] = np.where(rng.random(len(k)) < 0.5, left[k, ax], left[k, 3 + ax])
    maxd = np.where(rng.random(n) < 0.5, np.float32(3.402823466e38), rng.random(n, dtype=np.float32) * scale * 3).astype(np.float32)
    hit, dist = p.api.probe_aabb(np.ascontiguousarray(boxes[:, 0:6]), rays, np.full(n, 3.402823466e38, np.float32))
    kk = np.where(hit[: n // 8] == 1)[0]
    maxd[kk] = dist[kk]
    out = p.api.probe_boxpair_certified(boxes, rays, maxd)
    reg = out[:, 0] == 1
    tot += n
    reg_tot += int(reg.sum())
    redo += int(out[reg, 1].sum())
    bad += int(((out[reg, 2] != out[reg, 5]) | (out[reg, 3] != out[reg, 6]) | (out[reg, 4] != out[reg, 7])).sum())
    if c % 8 == 7:
        print(f"... {tot} cases, {reg_tot} in the class, {redo} exact redos, {bad} differing decisions", flush=True)
print(f"DONE: {tot} box pairs, {reg_tot} in the fast-division class, {redo} took the exact redo ({redo / max(1, reg_tot):.3f}), {bad} decisions differ from aabb::intersects")
sys.exit(1 if bad else 0)

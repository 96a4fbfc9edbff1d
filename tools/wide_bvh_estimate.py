#!/usr/bin/env python3
"""Would a 4-wide BVH pay on this kernel?  An estimate from visit counts (dev tool, CPU only, numpy).

Takes the flat binary BVH the product builds for the Book-1 final scene (rt_scene_get_flat), collapses it to a 4-ary tree (every wide node
adopts its grandchildren; leaves stay), and walks both trees with the reference's traversal rule (test the children's boxes, nearest first,
cull at push time with the current closest hit) for a few thousand rays: primary rays of the camera and rays scattered from random points
on the spheres.  Counts, per ray: node visits, box tests, sphere tests.  The instruction cost per visit is what the ISA of the streaming
kernel shows for the binary visit (82 vector + 21 scalar instructions) and what the same code shape needs for four boxes
(24 exact quotients, a 4-element sorting network, up to three pushes).  Plain fp32 numpy: counts, not bit-exact results.

    python tools/wide_bvh_estimate.py [--rays 4000]
"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G

ap = argparse.ArgumentParser()
ap.add_argument("--rays", type=int, default=4000)
a = ap.parse_args()
p = G.load_package()
scene = p.Scene.book1_final(1984)
nodes, prims, _ = scene.arrays()
root = scene.getWorldPtr().root
rng = np.random.default_rng(1)

# ---- rays: half primary (defocus camera of config 2, pixel centres), half scattered from sphere surfaces --------------------------
cam = p.DefocusBlurCamera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, 1.5, 0.1, 10.0)
o = np.array(cam.o, np.float32); u = np.array(cam.u, np.float32); v = np.array(cam.v, np.float32); w = np.array(cam.w, np.float32)
n1 = a.rays // 2
st = rng.random((n1, 2), dtype=np.float32) * 2 - 1
d1 = (w * cam.focus_dist)[None] + (u * cam.viewport_width * cam.focus_dist)[None] * st[:, :1] + (v * cam.viewport_height * cam.focus_dist)[None] * st[:, 1:]
rays = [(o.copy(), d1[i]) for i in range(n1)]
for _ in range(a.rays - n1):
    k = int(rng.integers(0, len(prims)))
    nrm = rng.standard_normal(3).astype(np.float32); nrm /= np.linalg.norm(nrm)
    pt = prims[k]["c0"] + nrm * prims[k]["radius"]
    dirn = nrm + (lambda x: x / np.linalg.norm(x))(rng.standard_normal(3).astype(np.float32))
    rays.append((pt + dirn * np.float32(0.001), dirn.astype(np.float32)))


def box(n, o, d, maxd):
    with np.errstate(divide="ignore", invalid="ignore"):
        t0 = (n["min"] - o) / d; t1 = (n["max"] - o) / d
    tmin = np.max(np.minimum(t0, t1)); tmax = np.min(np.maximum(t0, t1))
    return (tmin <= tmax and tmin < maxd and tmax > 0), tmin


def sphere(pr, o, d):
    oc = o - pr["c0"]; aa = d @ d; hb = d @ oc; c = oc @ oc - pr["radius"] ** 2
    disc = hb * hb - aa * c
    if disc <= 0: return np.inf
    s = np.sqrt(disc); t = (-hb - s) / aa
    if t < 0:
        t = (-hb + s) / aa
        if t < 0: return np.inf
    return t


def children4(i):
    """the (up to 4) children a collapsed node adopts: grandchildren where the child is inner, the child itself where it is a leaf"""
    out = []
    for c in (nodes[i]["left"], nodes[i]["right"]):
        if nodes[c]["left"] == -1: out.append(c)
        else: out += [nodes[c]["left"], nodes[c]["right"]]
    return out


def walk(o, d, wide):
    best, visits, boxes, leaves = np.inf, 0, 0, 0
    ok, _ = box(nodes[root], o, d, best); boxes += 1
    stack = [root] if ok else []
    while stack:
        i = stack.pop()
        if nodes[i]["left"] == -1:
            leaves += 1
            t = sphere(prims[nodes[i]["right"]], o, d)
            if t < best: best = t
            continue
        visits += 1
        kids = children4(i) if wide else [nodes[i]["left"], nodes[i]["right"]]
        hit = []
        for c in kids:
            ok, tm = box(nodes[c], o, d, best); boxes += 1
            if ok: hit.append((tm, c))
        for tm, c in sorted(hit, key=lambda x: -x[0]):   # far ones first onto the stack, the nearest is popped next
            stack.append(c)
    return visits, boxes, leaves


res = {}
for wide in (False, True):
    tot = np.zeros(3)
    for o_, d_ in rays:
        tot += walk(o_, d_, wide)
    res[wide] = tot / len(rays)
    print(f"{'4-wide' if wide else 'binary'}: {res[wide][0]:.2f} node visits, {res[wide][1]:.2f} box tests, {res[wide][2]:.2f} sphere tests per ray")
b, w4 = res[False], res[True]
cost_b, cost_w = 82 + 21, 24 * 4 + 24 + 8 + 12 + 25 + 12 + 30   # vector + scalar per visit (binary: from the ISA; 4-wide: quotients, subs, max3/min3, compares, sort, pushes, scalar)
print(f"instructions per ray spent in node visits: binary {b[0] * cost_b:.0f}, 4-wide {w4[0] * cost_w:.0f}  ({cost_b} / {cost_w} per visit)")

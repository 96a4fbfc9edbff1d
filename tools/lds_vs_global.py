#!/usr/bin/env python3
"""Mid-size worlds: LDS-resident image with ONE workgroup per CU against the global-memory path with two.
    python tools/lds_vs_global.py   (run twice: plain and with RT06_FORCE_BIG=1)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
p = G.load_package()
W, H, spp = 800, 600, 64
for n in (400, 700, 1100, 1400):
    rng = np.random.default_rng(n)
    s = p.Scene()
    mats = [s.Lambertian((0.7, 0.3, 0.3)), s.Metal((0.8, 0.8, 0.9), 0.1), s.Dielectric((1, 1, 1), 1.5), s.Lambertian((0.3, 0.3, 0.8))]
    s.MakeSphere((0, -1000.0, 0), 1000.0, mats[3])
    for i in range(n):
        c = ((rng.random(3) * 2 - 1) * np.array([12, 0, 12]) + np.array([0, 0.2 + rng.random() * 1.5, 0])).astype(np.float32)
        s.MakeSphere(c, float(0.05 + rng.random() * 0.25), mats[int(rng.integers(0, 4))])
    s.BuildBVH_TopDown()
    cam = p.DefocusBlurCamera((13, 2, 3), (0, 0, 0), (0, 1, 0), 25.0, W / H, 0.05, 10.0)
    r = p.Renderer.MakeRenderer(W, H, spp, 50, cam, s.getWorldPtr())
    best = 1e9
    for _ in range(3):
        r.Render(); best = min(best, r.last_kernel_ms())
    print(n, r.kernel_info(), f"{best:.2f} ms  {W*H*spp/best/1e3:.0f} Msamples/s", flush=True)
    r.close()

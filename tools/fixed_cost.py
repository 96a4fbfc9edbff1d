#!/usr/bin/env python3
"""Per-launch fixed cost of the streaming kernel: render time against spp (best of 4), for two depth limits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
p = G.load_package()
W, H = 1200, 800
scene = p.Scene.book1_final(1984); cam = p.DefocusBlurCamera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, W / H, 0.1, 10.0)
for depth in (50, 4):
    row = []
    for spp in (1, 2, 4, 8, 16, 32, 64):
        r = p.Renderer.MakeRenderer(W, H, spp, depth, cam, scene.getWorldPtr())
        best = 1e9
        for _ in range(4):
            r.Render(); best = min(best, r.last_kernel_ms())
        r.close()
        row.append((spp, round(best, 3)))
    slope = (row[-1][1] - row[-2][1]) / 32.0
    print(f"depth {depth}: {row}  slope {slope:.4f} ms/spp, intercept {row[-1][1] - 64 * slope:.3f} ms", flush=True)

#!/usr/bin/env python3
"""Where the per-launch fixed cost sits: per-kernel HIP-event times of small renders of BASELINE configs[1] (best of 5)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
p = G.load_package()
W, H = 1200, 800
scene = p.Scene.book1_final(1984); cam = p.DefocusBlurCamera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, W / H, 0.1, 10.0)
for ws in (1, 8):
    for depth in (50, 4, 1):
        for spp in (1, 4, 16, 64):
            r = p.Renderer.MakeRenderer(W, H, spp, depth, cam, scene.getWorldPtr(), rank=0, world_size=ws)
            best = None
            for _ in range(5):
                r.Render()
                t = r.kernel_times() + (r.last_kernel_ms(),)
                if best is None or t[3] < best[3]:
                    best = t
            r.close()
            print(f"world_size {ws} depth {depth:2d} spp {spp:3d}: primary {best[0]:.3f}  stream {best[1]:.3f}  resolve {best[2]:.3f}  all {best[3]:.3f} ms", flush=True)

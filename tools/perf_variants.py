#!/usr/bin/env python3
"""Quick variant timing on BASELINE.json configs[1] (dev tool): python tools/perf_variants.py [variants...]
env: W H SPP DEPTH SCENE"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
p = G.load_package()
W, H, spp = int(os.environ.get("W", 1200)), int(os.environ.get("H", 800)), int(os.environ.get("SPP", 500))
depth = int(os.environ.get("DEPTH", 50))
which = os.environ.get("SCENE", "book1_final")
scene = getattr(p.Scene, which)(1984)
cam = p.DefocusBlurCamera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, W / H, 0.1, 10.0) if which == "book1_final" else p.MotionBlurCamera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, W / H, 0.0, 1.0)
for v in [int(a) for a in sys.argv[1:]] or [1, 2]:
    r = p.Renderer.MakeRenderer(W, H, spp, depth, cam, scene.getWorldPtr(), variant=v)
    best = 1e30
    for it in range(3):
        r.Render()
        best = min(best, r.last_kernel_ms())
    print(f"variant {v} depth {depth}: {best:.2f} ms  {W*H*spp/best/1e3:.1f} Msamples/s", flush=True)
    r.close()

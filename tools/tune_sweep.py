#!/usr/bin/env python3
"""Sweep the wave-scheduling thresholds of render_kernel_stream (RT06_TUNE=keep,shade,leaf) in one process (dev tool).
    python tools/tune_sweep.py 40,56,4 32,48,4 ...      env: W H SPP DEPTH SCENE REPS"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
p = G.load_package()
W, H, spp = int(os.environ.get("W", 1200)), int(os.environ.get("H", 800)), int(os.environ.get("SPP", 500))
depth = int(os.environ.get("DEPTH", 50))
reps = int(os.environ.get("REPS", 3))
which = os.environ.get("SCENE", "book1_final")
scene = getattr(p.Scene, which)(1984) if which != "cornell_box" else p.Scene.cornell_box()
if which == "book1_final":
    cam = p.DefocusBlurCamera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, W / H, 0.1, 10.0)
elif which == "cornell_box":
    cam = p.PinholeCamera((278, 278, -800), (278, 278, 0), (0, 1, 0), 40.0, W / H)
elif which == "book2_final":
    cam = p.MotionBlurCamera((478, 278, -600), (278, 278, 0), (0, 1, 0), 40.0, W / H, 0.0, 1.0)
else:
    cam = p.MotionBlurCamera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, W / H, 0.0, 1.0)
for t in sys.argv[1:] or ["40,56,4"]:
    os.environ["RT06_TUNE"] = t
    r = p.Renderer.MakeRenderer(W, H, spp, depth, cam, scene.getWorldPtr())
    best = 1e30
    for it in range(reps):
        r.Render()
        best = min(best, r.last_kernel_ms())
    print(f"tune {t}: {best:.2f} ms  {W*H*spp/best/1e3:.1f} Msamples/s", flush=True)
    r.close()

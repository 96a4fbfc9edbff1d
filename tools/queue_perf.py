#!/usr/bin/env python3
"""The alternative traversal rules on BASELINE configs[1] — the reference's disabled distance-sorted queue (BVH.cu:17-49) and the 4-wide walk of the
same tree (RT_TRAVERSAL_WIDE4) — in the streaming kernel's lane-walk mode and on the baseline kernel, against the live stack walk.
python tools/queue_perf.py [spp]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
p = G.load_package()
W, H, spp = 1200, 800, int(sys.argv[1]) if len(sys.argv) > 1 else 64
cam = p.DefocusBlurCamera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, W / H, 0.1, 10.0)
for label, traversal, variant in (("stack walk, streaming kernel (variant 3)", 0, 0), ("queue, streaming kernel (queue mode of variant 2)", 1, 0), ("queue, baseline kernel (variant 1)", 1, 1),
                                 ("4-wide walk, streaming kernel (lane-walk mode)", 2, 0), ("4-wide walk, baseline kernel (variant 1)", 2, 1), ("stack walk, baseline kernel (variant 1)", 0, 1),
                                 ("stack walk, streaming kernel variant 2 (IEEE divisions)", 0, 2)):
    s = p.Scene.book1_final(1984).set_traversal(traversal)
    r = p.Renderer.MakeRenderer(W, H, spp, 50, cam, s.getWorldPtr(), variant=variant)
    best = 1e30
    for _ in range(3):
        r.Render(); best = min(best, r.last_kernel_ms())
    print(f"{label:52s}: {best:8.2f} ms  {W * H * spp / best / 1e3:7.1f} Msamples/s  {r.kernel_info()}", flush=True)
    r.close()

#!/usr/bin/env python3
"""The program rocprofv3 profiles for the exchange kernel: N renders of BASELINE configs[1] with kernel variant VARIANT (default 5).
    rocprofv3 --kernel-trace --pmc ... -- python3 tools/xchg_prof.py [variant] [renders]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
p = G.load_package()
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
W, H, spp = 1200, 800, 500
scene = p.Scene.book1_final(1984); cam = p.DefocusBlurCamera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, W / H, 0.1, 10.0)
r = p.Renderer.MakeRenderer(W, H, spp, 50, cam, scene.getWorldPtr(), variant=variant)
for _ in range(n):
    r.Render()
print("kernel ms", r.kernel_times(), flush=True)
r.close()

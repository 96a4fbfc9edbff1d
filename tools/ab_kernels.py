#!/usr/bin/env python3
"""A/B of library builds on ONE box by per-kernel HIP-event times (BASELINE configs[1], or configs[4]'s scene with AB_WORKLOAD=book2_final; best and mean of N renders):
    python tools/ab_kernels.py libA.so libB.so ...   (each library is loaded in its own child process; two rounds)"""
import os, subprocess, sys
CHILD = r'''
import os, sys
sys.path.insert(0, os.getcwd())
import __graft_entry__ as G
import numpy as np
p = G.load_package()
if os.environ.get("AB_WORKLOAD") == "book2_final":   # BASELINE configs[4]'s scene: the global-memory form of the kernel
    W, H, spp = 1920, 1080, 64
    scene = p.Scene.book2_final(1984); cam = p.MotionBlurCamera((478, 278, -600), (278, 278, 0), (0, 1, 0), 40.0, W / H, 0.0, 1.0)
elif os.environ.get("AB_WORKLOAD") == "cornell_box":   # BASELINE configs[3]
    W, H, spp = 600, 600, 1000
    scene = p.Scene.cornell_box(); cam = p.PinholeCamera((278, 278, -800), (278, 278, 0), (0, 1, 0), 40.0, W / H)
else:
    W, H, spp = 1200, 800, 500
    scene = p.Scene.book1_final(1984); cam = p.DefocusBlurCamera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, W / H, 0.1, 10.0)
r = p.Renderer.MakeRenderer(W, H, spp, 50, cam, scene.getWorldPtr())
r.Render()
t = []
for _ in range(8):
    r.Render(); t.append(r.kernel_times() + (r.last_kernel_ms(),))
t = np.array(t)
print("  mean primary %.2f dominant %.2f resolve %.2f all %.2f | best all %.2f ms" % (t[:, 0].mean(), t[:, 1].mean(), t[:, 2].mean(), t[:, 3].mean(), t[:, 3].min()), flush=True)
'''
for rnd in (1, 2):
    for lib in sys.argv[1:]:
        print(lib, flush=True)
        env = dict(os.environ, RT06_LIB=os.path.abspath(lib))
        subprocess.run([sys.executable, "-c", CHILD], env=env, check=True)

#!/usr/bin/env python3
"""One big-frame render of the exchange kernel with diagnostics (RT06_DEBUG=1): python tools/xchg_debug.py [spp] [setting]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RT06_DEBUG"] = "1"
if len(sys.argv) > 2:
    os.environ["RT06_XCHG"] = sys.argv[2]
import __graft_entry__ as G
p = G.load_package()
W, H, spp = 1200, 800, int(sys.argv[1]) if len(sys.argv) > 1 else 16
scene = p.Scene.book1_final(1984); cam = p.DefocusBlurCamera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, W / H, 0.1, 10.0)
r3 = p.Renderer.MakeRenderer(W, H, spp, 50, cam, scene.getWorldPtr(), variant=3)
r3.Render(); ref = r3.DownloadRenderbuffer(); r3.close()
r = p.Renderer.MakeRenderer(W, H, spp, 50, cam, scene.getWorldPtr(), variant=5)
for k in range(3):
    t = time.time()
    try:
        r.Render()
        img = r.DownloadRenderbuffer()
        print(f"render {k}: {time.time() - t:.3f} s, kernel {r.kernel_times()[1]:.2f} ms, same bits as variant 3: {img.tobytes() == ref.tobytes()}", flush=True)
    except Exception as e:
        print(f"render {k}: {time.time() - t:.3f} s FAILED: {e}", flush=True)
        break
r.close()

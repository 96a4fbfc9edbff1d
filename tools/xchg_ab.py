#!/usr/bin/env python3
"""A/B on one box: the streaming kernel (variant 3) against the ray-exchange kernel (variant 5) under RT06_XCHG settings.
    python tools/xchg_ab.py [workload] [setting ...]      setting = tracers,extra,swap,shade,patience,prio,keep
Prints the per-kernel HIP-event times (best of 3 renders) per setting."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
p = G.load_package()
wl = sys.argv[1] if len(sys.argv) > 1 else "book1_final"
settings = sys.argv[2:] or ["8,192,16,48,6,1,44"]
if wl == "book1_final":
    W, H, spp = 1200, 800, 500
    scene = p.Scene.book1_final(1984); cam = p.DefocusBlurCamera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, W / H, 0.1, 10.0)
else:
    W, H, spp = 800, 800, 1000
    scene = p.Scene.book2_moving(1984); cam = p.MotionBlurCamera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, W / H, 0.0, 1.0)


def run(variant, label):
    r = p.Renderer.MakeRenderer(W, H, spp, 50, cam, scene.getWorldPtr(), variant=variant)
    best = None
    for _ in range(3):
        r.Render()
        t = r.kernel_times()
        if best is None or t[1] < best[1]:
            best = t
    img = r.DownloadRenderbuffer()
    r.close()
    print(f"{label:40s} primary {best[0]:6.2f}  dominant {best[1]:7.2f}  resolve {best[2]:5.2f}  sum {sum(best):7.2f} ms", flush=True)
    return img


ref = run(3, "variant 3 (streaming)")
for s in settings:
    os.environ["RT06_XCHG"] = s
    img = run(5, f"variant 5 RT06_XCHG={s}")
    print("    same bits as variant 3:", img.tobytes() == ref.tobytes(), flush=True)
run(3, "variant 3 (streaming) again")

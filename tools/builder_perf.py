#!/usr/bin/env python3
"""Render BASELINE.json configs[1] with each of the reference's three BVH builders (same image bits, different trees):
median split (_build_bvh_rec1, the live default), binned SAH (_build_bvh_rec2, disabled in the reference), bottom-up."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
p = G.load_package()
W, H, spp = 1200, 800, int(os.environ.get("SPP", 500))
cam = p.DefocusBlurCamera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, W / H, 0.1, 10.0)
ref = None
for name in ("BuildBVH_TopDown", "BuildBVH_SAH", "BuildBVH_BottomUp"):
    s = p.Scene.book1_final(1984)
    getattr(s, name)()
    w = s.getWorldPtr()
    r = p.Renderer.MakeRenderer(W, H, spp, 50, cam, w)
    best = 1e9
    for _ in range(3):
        r.Render(); best = min(best, r.last_kernel_ms())
    img = r.DownloadRenderbuffer()
    same = "-" if ref is None else str(img.tobytes() == ref.tobytes())
    ref = img if ref is None else ref
    print(f"{name:18s} nodes {w.n_nodes} max_stack {w.max_stack}: {best:.2f} ms  {W*H*spp/best/1e3:.1f} Msamples/s  same image bits as median-split: {same}", flush=True)
    r.close()
import numpy as np
s1 = p.Scene.book1_final(1984); s2 = p.Scene.book1_final(1984); s2.BuildBVH_SAH()
imgs = []
for s in (s1, s2):
    r = p.Renderer.MakeRenderer(W, H, spp, 50, cam, s.getWorldPtr()); r.Render(); imgs.append(r.DownloadRenderbuffer()); r.close()
d = np.abs(imgs[0][..., :3] - imgs[1][..., :3]).max(axis=2)
print(f"median-split vs SAH image: {int((d > 0).sum())} of {W*H} pixels differ, max |delta| {np.nanmax(d):.3e}, mean over differing {d[d>0].mean():.3e}")

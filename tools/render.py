#!/usr/bin/env python3
"""Command-line caller of the path — what FirstApp::MakeApp / Run hard-code (main/src/FirstApp.cpp:20-56,94-106):

    python tools/render.py --scene book1_final --width 1200 --height 800 --spp 500 --depth 50 --out out.png

Scenes: book1_final (DefocusBlurCamera vfov 20, aperture 0.1), book2_moving (MotionBlurCamera t in [0,1]),
three_spheres (PinholeCamera vfov 90).  Prints one JSON line with the render time and Msamples/s.
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="book1_final", choices=["book1_final", "book2_moving", "three_spheres", "cornell_box", "book2_final"])
ap.add_argument("--width", type=int, default=1200)
ap.add_argument("--height", type=int, default=800)
ap.add_argument("--spp", type=int, default=500)
ap.add_argument("--depth", type=int, default=50)
ap.add_argument("--seed", type=int, default=1984)
ap.add_argument("--device", type=int, default=0)
ap.add_argument("--gpus", type=int, default=1, help="> 1: tile-shard the frame over GPUs 0..N-1 of this node from this one process (rt_multi_renderer_*, one RCCL exchange)")
ap.add_argument("--out", default="render.png")
a = ap.parse_args()
p = G.load_package()
from ray_tracing_v06_amd import image_io
W, H = a.width, a.height
if a.scene == "three_spheres":
    scene, cam = p.Scene.three_spheres(), p.PinholeCamera((0, 0, 0), (0, 0, -1), (0, 1, 0), 90.0, W / H)
elif a.scene == "cornell_box":
    scene, cam = p.Scene.cornell_box(), p.PinholeCamera((278, 278, -800), (278, 278, 0), (0, 1, 0), 40.0, W / H)
elif a.scene == "book2_final":
    scene, cam = p.Scene.book2_final(a.seed), p.MotionBlurCamera((478, 278, -600), (278, 278, 0), (0, 1, 0), 40.0, W / H, 0.0, 1.0)
elif a.scene == "book1_final":
    scene, cam = p.Scene.book1_final(a.seed), p.DefocusBlurCamera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, W / H, 0.1, 10.0)
else:
    scene, cam = p.Scene.book2_moving(a.seed), p.MotionBlurCamera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, W / H, 0.0, 1.0)
if a.gpus > 1:
    r = p.MultiRenderer.MakeRenderer(W, H, a.spp, a.depth, cam, scene.getWorldPtr(), a.gpus, seed=a.seed)
    r.Render()
    ms = r.times()[0]     # host wall-clock of Render(): all shards, the exchange, the assembly
else:
    r = p.Renderer.MakeRenderer(W, H, a.spp, a.depth, cam, scene.getWorldPtr(), seed=a.seed, device=a.device)
    r.Render()
    ms = r.last_kernel_ms()
fb = r.DownloadRenderbuffer()
# .jpg = the reference app's own format (stbi_write_jpg quality 95, FirstApp.cpp:120); .ppm / .png are lossless
(image_io.write_ppm if a.out.endswith(".ppm") else image_io.write_jpg if a.out.endswith((".jpg", ".jpeg")) else image_io.write_png)(a.out, fb)
print(json.dumps({"scene": a.scene, "width": W, "height": H, "spp": a.spp, "max_depth": a.depth, "render_ms": round(ms, 3),
                  "msamples_per_s": round(W * H * a.spp / ms / 1e3, 1), "gpus": a.gpus, "out": a.out}))

import os, sys
sys.path.insert(0, os.getcwd())
import __graft_entry__ as G
p = G.load_package()
W, H, spp = 1920, 1080, 16
scene = p.Scene.book2_final(1984); cam = p.MotionBlurCamera((478, 278, -600), (278, 278, 0), (0, 1, 0), 40.0, W / H, 0.0, 1.0)
r = p.Renderer.MakeRenderer(W, H, spp, 50, cam, scene.getWorldPtr())
r.Render()
sys.stderr.flush()
print(r.kernel_times(), file=sys.stderr)
r.close()

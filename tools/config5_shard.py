import sys, os
sys.path.insert(0, os.getcwd())
import __graft_entry__ as G
p = G.load_package()
W,H,spp=3840,2160,64
s=p.Scene.book2_final(1984); cam=p.MotionBlurCamera((478,278,-600),(278,278,0),(0,1,0),40.0,W/H,0.0,1.0)
for ws in (8,1):
    r=p.Renderer.MakeRenderer(W,H,spp,40,cam,s.getWorldPtr(),rank=0,world_size=ws)
    r.Render(); ms=r.last_kernel_ms()
    print(f"config5 3840x2160x{spp} depth 40, world_size {ws} rank 0: {ms:.1f} ms -> {W*H*spp/ws/ms/1e3:.0f} Msamples/s per GPU", flush=True)
    r.close()

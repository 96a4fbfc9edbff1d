import sys, os
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/__graft_entry__.py") else os.getcwd())
import __graft_entry__ as G
p = G.load_package()
W,H,spp=1200,800,500
scene=p.Scene.book1_final(1984); cam=p.DefocusBlurCamera((13,2,3),(0,0,0),(0,1,0),20.0,W/H,0.1,10.0)
for ws in (1,2,4,8,16):
    r=p.Renderer.MakeRenderer(W,H,spp,50,cam,scene.getWorldPtr(),rank=0,world_size=ws)
    best=1e9
    for i in range(4):
        r.Render(); best=min(best,r.last_kernel_ms())
    print(f"world_size {ws}: rank0 {best:.2f} ms -> {W*H*spp/ws/best/1e3:.1f} Msamples/s per GPU", flush=True)
    r.close()

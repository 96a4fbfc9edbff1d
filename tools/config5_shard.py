#!/usr/bin/env python3
"""BASELINE configs[4] (Book-2 final scene, 3840x2160, depth 50) as ONE rank of an 8-GPU tile-sharded run sees it: rank 0 of 8 renders
its 1/8 shard at the given spp (default 10 000, the configuration's own) on this GPU.  One JSON line.
    python tools/config5_shard.py [spp] [world_size]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
p = G.load_package()
W, H, depth = 3840, 2160, 50
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
ws = int(sys.argv[2]) if len(sys.argv) > 2 else 8
s = p.Scene.book2_final(1984)
cam = p.MotionBlurCamera((478, 278, -600), (278, 278, 0), (0, 1, 0), 40.0, W / H, 0.0, 1.0)
r = p.Renderer.MakeRenderer(W, H, spp, depth, cam, s.getWorldPtr(), rank=0, world_size=ws)
pi = r.pass_info()
r.Render()
ms = r.last_kernel_ms()
kt = r.kernel_times()
print(json.dumps({"workload": f"book2_final {W}x{H}x{spp} depth {depth}", "world_size": ws, "rank": 0, "ms": round(ms, 1),
                  "Msamples_per_s_this_gpu": round(W * H * spp / ws / ms / 1e3, 1), "passes": pi["n_passes"], "spp_per_pass": pi["pass_spp"],
                  "kernel_ms_sum_over_passes": {"primary_rays": round(kt[0], 1), "dominant": round(kt[1], 1), "resolve": round(kt[2], 1)},
                  "kernel": r.kernel_info()}), flush=True)
r.close()

#!/usr/bin/env python3
"""Does keeping two frames in flight hide the tail of a shard?  Rank 0 of an N-way tile shard of BASELINE configs[1] rendered K times
(a) back to back on one HIP stream, (b) alternating between two renderers on two HIP streams (bench.py --pipeline 2 without the
collectives).  One GPU; steady-state ms per frame.
    python tools/shard_pipeline.py [world_size] [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as G
p = G.load_package()
ws = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
W, H, spp = 1200, 800, 500
scene = p.Scene.book1_final(1984); cam = p.DefocusBlurCamera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, W / H, 0.1, 10.0)
for depth in (1, 2):
    rs = [p.Renderer.MakeRenderer(W, H, spp, 50, cam, scene.getWorldPtr(), rank=0, world_size=ws) for _ in range(depth)]
    bufs = [torch.zeros(r.shard_floats(), dtype=torch.float32, device="cuda:0") for r in rs]
    streams = [torch.cuda.Stream() for _ in range(depth)]
    for k in range(4):
        rs[k % depth].render_async(streams[k % depth].cuda_stream, bufs[k % depth].data_ptr())
    torch.cuda.synchronize()
    t = time.perf_counter()
    for k in range(K):
        rs[k % depth].render_async(streams[k % depth].cuda_stream, bufs[k % depth].data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / K * 1e3
    print(f"world_size {ws} rank 0, {depth} frame(s) in flight: {dt:.3f} ms per frame ({W * H * spp / ws / dt / 1e3:.0f} Msamples/s)", flush=True)
    for r in rs:
        r.close()

#!/usr/bin/env python3
"""Phase shares of the ray-exchange kernel (a -DRT_PHASE_TIMERS build, loaded with RT06_LIB): one render per RT06_XCHG setting.
    RT06_LIB=$PWD/ray-tracing-v06_amd/csrc/build/librt06_timers.so python tools/xchg_phase.py [setting ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
p = G.load_package()
W, H, spp = 1200, 800, int(os.environ.get("SPP", "200"))
scene = p.Scene.book1_final(1984); cam = p.DefocusBlurCamera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, W / H, 0.1, 10.0)
for s in sys.argv[1:] or ["8,192,16,48,6,1,44"]:
    variant = 3 if s == "v3" else 5
    os.environ["RT06_XCHG"] = s
    r = p.Renderer.MakeRenderer(W, H, spp, 50, cam, scene.getWorldPtr(), variant=variant)
    r.Render()
    sys.stderr.flush()
    print(f"== {s}: {r.kernel_times()[1]:.2f} ms for {spp} spp", file=sys.stderr, flush=True)
    r.close()

#!/usr/bin/env python3
"""bench.py — Msamples/s of the per-pixel sample loop on the Book-1 final scene (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one Render() of the whole 1200x800x500-spp frame (4.8e8 pixel-samples) with the flat scene
already resident in HBM: N == 1 launches the kernel(s) into the row-major framebuffer; N > 1 tile-shards
the frame over the ranks (one process per GPU), each rank renders its shard, ONE RCCL gather moves the
shards to rank 0 over xGMI and a de-interleave kernel assembles the image.  Total work is fixed as N
grows ("strong" scaling).  Rank 0 prints one JSON line.

`roofline` prices the dominant kernel against HBM with SURVEY.md §8(d)'s ALGORITHMIC bytes per sample
(32 B per box test + 16 B per sphere test + 16 B per shaded hit + 16/spp B of framebuffer), the counts
measured by the instrumented CPU oracle on the same scene and seed.  `cpu_baseline` times that oracle (a
CPU port of the loop — the reference has no CPU renderer) on the host cores, on a bounded spp.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


# BASELINE.json configs: the headline (default) is configs[1]; the others are parity-test cases that can be timed with
# the same harness.  (product scene ctor, camera ctor + args, oracle scene ctor, oracle camera ctor, W, H, spp, depth, text)
WORKLOADS = {
    "book1_final": ("book1_final", "DefocusBlurCamera", ((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, None, 0.1, 10.0), "camera_defocus",
                    1200, 800, 500, 50, "Book-1 final random-spheres scene (488 spheres, 975-node BVH)", "DefocusBlurCamera vfov 20 aperture 0.1"),
    "book2_moving": ("book2_moving", "MotionBlurCamera", ((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, None, 0.0, 1.0), "camera_motion",
                     800, 800, 1000, 50, "Book-2 moving-spheres scene (488 spheres, 975-node BVH)", "MotionBlurCamera vfov 20 shutter 0..1"),
    "cornell_box": ("cornell_box", "PinholeCamera", ((278, 278, -800), (278, 278, 0), (0, 1, 0), 40.0, None), "camera_pinhole",
                    600, 600, 5000, 50, "Cornell box of The Next Week (18 quads, area light; not in the reference)", "PinholeCamera vfov 40"),
    "book2_final": ("book2_final", "MotionBlurCamera", ((478, 278, -600), (278, 278, 0), (0, 1, 0), 40.0, None, 0.0, 1.0), "camera_motion",
                    800, 800, 1000, 40, "final scene of The Next Week (2401 quads, 1008 spheres, media, Perlin, image texture; not in the reference)",
                    "MotionBlurCamera vfov 40 shutter 0..1"),
}


def make_workload(args, mod, oracle=False):
    """(scene, camera) of the selected workload through the product package (oracle=False) or the CPU oracle bindings."""
    wl = WORKLOADS[args.workload]
    cam_args = tuple(args.width / args.height if a is None else a for a in wl[2])
    if oracle:
        scene = getattr(mod.Scene, wl[0])() if wl[0] == "cornell_box" else getattr(mod.Scene, wl[0])(args.seed)
        return scene, getattr(mod, wl[3])(*cam_args)
    scene = getattr(mod.Scene, wl[0])() if wl[0] == "cornell_box" else getattr(mod.Scene, wl[0])(args.seed)
    return scene, getattr(mod, wl[1])(*cam_args)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="book1_final", choices=sorted(WORKLOADS),
                    help="book1_final = BASELINE configs[1], the headline (default); the others time configs[2..4] with the same harness")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=None, help="default: the workload's (1200 for the headline)")
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--depth", type=int, default=None)
    ap.add_argument("--seed", type=int, default=1984)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target duration of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--pipeline", type=int, default=0, choices=[0, 1, 2],
                    help="frames in flight: 2 renders frame k+1 on a second HIP stream while frame k drains its last paths and is gathered "
                         "(two renderers, two shard buffers; the collectives stay on one stream, in order); 1 = strictly serial; "
                         "0 (default) = 1.  Measured: no gain on one GPU (a persistent workgroup frees its LDS only when its last wave "
                         "ends, so the next frame cannot move in early); on several GPUs it hides the gather behind the next render")
    ap.add_argument("--verify-assembly", action="store_true",
                    help="N > 1: after the timed region rank 0 renders the frame alone and requires the assembled image to be the same bits")
    a = ap.parse_args()
    wl = WORKLOADS[a.workload]
    a.width = a.width or wl[4]
    a.height = a.height or wl[5]
    a.spp = a.spp or wl[6]
    a.depth = a.depth or wl[7]
    return a


def host_cores():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def profiled_traffic(kernel_substr):
    """HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 PMC summary
    (profiles/rNN_bench_pmc_summary.csv: separate --pmc FETCH_SIZE / WRITE_SIZE passes of this same command).
    MI355X_MICROARCH.md §HBM: both counters are in KB; on gfx950 FETCH_SIZE counts 64 B per 128-B request on
    wide coalesced reads, so it is doubled.  Returns (bytes, source) or (None, None)."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_pmc_summary.csv")))
    if not files:
        return None, None
    vals = {}
    for row in csv.DictReader(open(files[-1])):
        if kernel_substr in row["kernel"] and row["counter"] in ("FETCH_SIZE", "WRITE_SIZE"):
            vals[row["counter"]] = float(row["mean_per_dispatch"])
    if len(vals) != 2:
        return None, None
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, os.path.relpath(files[-1], ROOT)


def profiled_valu(kernel_substr):
    """What actually bounds the kernel (VALU issue), from the same committed PMC summary: wave-instructions per
    launch, fraction of the VALU issue peak (one wave64 instruction per 2 cycles per SIMD, 1024 SIMDs, at the clock
    GRBM_GUI_ACTIVE / 8 XCDs measured for that launch) and mean fraction of the 64 lanes active per instruction."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_pmc_summary.csv")))
    if not files:
        return None
    v = {}
    for row in csv.DictReader(open(files[-1])):
        if kernel_substr in row["kernel"]:
            v[row["counter"]] = float(row["mean_per_dispatch"])
    need = ("SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "GRBM_GUI_ACTIVE")
    if any(k not in v for k in need):
        return None
    cycles = v["GRBM_GUI_ACTIVE"] / 8.0
    out = {"wave_instructions_per_launch": v["SQ_INSTS_VALU"],
           "issue_frac_of_peak": round(v["SQ_INSTS_VALU"] / (1024.0 * cycles * 0.5), 4),
           "lanes_active_frac": round(v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_INSTS_VALU"] * 64.0), 4)}
    if "SQ_INSTS_SALU" in v:
        out["scalar_instructions_per_launch"] = v["SQ_INSTS_SALU"]
    return out


def count_leg(args):
    """Per-sample box / sphere / hit counts of the algorithmic-bytes model, from a 1-spp oracle pass (N > 1 runs,
    where the timed cpu_baseline leg is skipped)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    W, H = args.width, args.height
    oscene, ocam = make_workload(args, O, oracle=True)
    _, cnt = O.render(oscene.world, ocam, W, H, 1, args.depth, args.seed, threads=host_cores())
    n = float(cnt.samples)
    return {"box_tests": cnt.box_tests / n, "leaf_tests": cnt.leaf_tests / n, "shaded_hits": cnt.shaded_hits / n,
            "rays": cnt.rays / n, "rng_draws": cnt.rng_draws / n}


def cpu_leg(args, pkg, scene, cam, gpu_image_fn):
    """cpu_baseline + algorithmic counts + parity sample, rank 0 at N == 1 only.  The oracle is the checker
    and the timed baseline here; it is never on the product path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    cores = host_cores()
    W, H = args.width, args.height
    oscene, ocam = make_workload(args, O, oracle=True)
    # calibrate on 1 spp, then size the sample for ~cpu_seconds of wall time
    t = time.perf_counter()
    O.render(oscene.world, ocam, W, H, 1, args.depth, args.seed, threads=cores)
    t1 = time.perf_counter() - t
    spp = int(max(1, min(args.spp, args.cpu_seconds / max(t1, 1e-3))))
    t = time.perf_counter()
    ref, cnt = O.render(oscene.world, ocam, W, H, spp, args.depth, args.seed, threads=cores)
    dt = time.perf_counter() - t
    n = float(cnt.samples)
    counts = {"box_tests": cnt.box_tests / n, "leaf_tests": cnt.leaf_tests / n, "shaded_hits": cnt.shaded_hits / n,
              "rays": cnt.rays / n, "rng_draws": cnt.rng_draws / n}
    base = {"value": round(n / dt / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"{W}x{H}x{spp}spp {args.workload}, depth {args.depth}, seed {args.seed}, {dt:.1f}s wall, "
                      f"gcc -O2 -ffp-contract=off, pthreads over rows"}
    parity = None
    if gpu_image_fn is not None:
        img = gpu_image_fn(spp)
        d = np.abs(img[..., :3] - ref[..., :3])
        parity = {"max_abs_delta": float(np.nanmax(d)), "mean_abs_delta": float(np.nanmean(d)),
                  "tolerance": 1e-3, "sample": f"{W}x{H}x{spp}spp GPU vs CPU oracle, same seed"}
    return base, counts, parity


def main():
    args = parse()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL on this pool; must precede any HIP call
    import torch
    import torch.distributed as dist

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world_size:
        if world_size == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world_size}")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world_size > 1:
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.backend)

    pkg = G.load_package()
    W, H, spp = args.width, args.height, args.spp
    scene, cam = make_workload(args, pkg)
    world = scene.getWorldPtr()
    # `depth` frames in flight: renderer / buffers / side stream d serve the frames k with k % depth == d
    depth = args.pipeline if args.pipeline else 1
    rs = [pkg.Renderer.MakeRenderer(W, H, spp, args.depth, cam, world, seed=args.seed, device=local_rank,
                                    rank=rank, world_size=world_size, variant=args.variant) for _ in range(depth)]
    r = rs[0]
    stream = torch.cuda.current_stream()
    side = [torch.cuda.Stream(device=dev) for _ in range(depth)] if depth > 1 else [stream]
    images = [torch.zeros(H * W * 4, dtype=torch.float32, device=dev) if rank == 0 else None for _ in range(depth)]
    image = images[0]
    from ray_tracing_v06_amd import multigpu
    shards = [torch.zeros(r.shard_floats(), dtype=torch.float32, device=dev) for _ in range(depth)] if world_size > 1 else None
    rendered = [torch.cuda.Event() for _ in range(depth)]
    consumed = [torch.cuda.Event() for _ in range(depth)]
    for e in consumed:
        e.record(stream)

    kernel_events = []
    frame = [0]

    def step(record):
        d = frame[0] % depth
        frame[0] += 1
        sd = side[d]
        sd.wait_event(consumed[d])            # the buffers of this slot are free again (frame k - depth was gathered / assembled)
        e0 = e1 = None
        if record:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(sd)
        target = images[d] if world_size == 1 else shards[d]
        rs[d].render_async(sd.cuda_stream, target.data_ptr())
        if record:
            e1.record(sd)
            kernel_events.append((e0, e1))
        rendered[d].record(sd)
        stream.wait_event(rendered[d])        # everything after the render stays on ONE stream, in frame order
        if world_size > 1:
            gathered = multigpu.gather_shards(shards[d], world_size, rank, dst=0)  # the single frame-end exchange (RCCL over xGMI)
            if rank == 0:
                rs[d].assemble(gathered.data_ptr(), images[d].data_ptr(), stream.cuda_stream)
        consumed[d].record(stream)

    def fence():
        if world_size > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    if world_size > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in kernel_events])) if kernel_events else float("nan")
    kernel_ms_source = "HIP events around every launch of the timed region"
    if depth > 1:
        # frames overlap inside the timed region, so an event pair there spans the wait for the previous frame's workgroups too:
        # take the kernel's duration from ONE more launch on the drained GPU (untimed, after the timed region)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        rs[0].render_async(stream.cuda_stream, (images[0] if world_size == 1 else shards[0]).data_ptr())
        e1.record(stream)
        torch.cuda.synchronize()
        kernel_ms = float(e0.elapsed_time(e1))
        kernel_ms_source = "one serial launch after the timed region (two frames are in flight inside it)"
        if world_size > 1:
            dist.barrier()

    total_samples = float(W) * H * spp
    out = None
    if rank == 0:
        base = counts = parity = None
        if world_size == 1 and args.cpu_seconds > 0:
            def gpu_image(spp_small):
                rr = pkg.Renderer.MakeRenderer(W, H, spp_small, args.depth, cam, world, seed=args.seed, device=local_rank,
                                               variant=args.variant)
                rr.Render()
                img = rr.DownloadRenderbuffer()
                rr.close()
                return img
            base, counts, parity = cpu_leg(args, pkg, scene, cam, None if args.no_parity else gpu_image)
        value = total_samples * args.steps / elapsed / 1e6
        out = {
            "metric": "Msamples/sec (WxHxspp) on Book-1 final scene" if args.workload == "book1_final" else f"Msamples/sec (WxHxspp) on {args.workload}",
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{WORKLOADS[args.workload][8]}, {W}x{H}, {spp} spp, "
                                   f"max_depth {args.depth}, seed {args.seed}, {WORKLOADS[args.workload][9]}",
                       "parallelism": f"tile-shard x{world_size} + 1 RCCL gather" if world_size > 1 else "single GPU",
                       "frames_in_flight": depth,
                       "kernel_variant": args.variant},
            "kernel_ms_per_step_rank0": round(kernel_ms, 3), "kernel_ms_source": kernel_ms_source,
        }
        if counts is None:
            counts = count_leg(args)
        if counts is not None:
            bytes_per_sample = 32.0 * counts["box_tests"] + 16.0 * counts["leaf_tests"] + 16.0 * counts["shaded_hits"] + 16.0 / spp
            launch_samples = total_samples / world_size
            achieved = bytes_per_sample * launch_samples / (kernel_ms * 1e-3) / 1e9
            default_cfg = (args.workload, W, H, spp, args.depth, args.variant, world_size) == ("book1_final", 1200, 800, 500, 50, 0, 1)
            traffic, traffic_src = profiled_traffic("render_kernel_stream") if default_cfg else (None, None)
            out["roofline"] = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                               "kernel": "render_kernel_stream", "algorithmic_bytes_per_sample": round(bytes_per_sample, 1),
                               "samples_per_launch": launch_samples, "kernel_ms": round(kernel_ms, 3),
                               "counts_per_sample": {k: round(v, 3) for k, v in counts.items()},
                               "traffic_source": traffic_src,
                               "valu": profiled_valu("render_kernel_stream") if default_cfg else None,
                               "note": "algorithmic bytes = BVH node / sphere / material records the traversal touches; they are "
                                       "served from the LDS-resident scene, not HBM, so frac can exceed 1 and measured HBM traffic "
                                       "(the 12-B-per-sample radiance buffer) is ~250x smaller: the kernel is instruction-issue bound (the peak of issue_frac_of_peak is the 2-cycle fp32 add/mul/fma rate; compares, selects, min/max cost 4 cycles and scalar instructions are not hidden) "
                                       "(DESIGN.md §7)"}
            if base is not None:
                out["cpu_baseline"] = base
            if parity is not None:
                out["parity"] = parity
        if args.verify_assembly and world_size > 1:
            solo = pkg.Renderer.MakeRenderer(W, H, spp, args.depth, cam, world, seed=args.seed, device=local_rank, variant=args.variant)
            solo.Render()
            ref = solo.DownloadRenderbuffer()
            solo.close()
            got = images[(frame[0] - 1) % depth].cpu().numpy().reshape(H, W, 4)   # the last frame rendered
            same = got.tobytes() == ref.tobytes()
            out["assembly_verified"] = bool(same)
            if not same:
                raise SystemExit("assembled multi-GPU image differs from the single-GPU image")
        print(json.dumps(out), flush=True)
    for rr in rs:
        rr.close()
    if world_size > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

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

`roofline` reports the resource that BINDS the dominant kernel (render_kernel_stream): vector-instruction issue.  The scene
is LDS-resident, so HBM is not the roof — `roofline.hbm` keeps SURVEY.md §8(d)'s algorithmic-bytes figure (32 B per box test
+ 16 B per sphere test + 16 B per shaded hit + 16/spp B, counts from the instrumented CPU oracle on the same scene and
seed) and the measured HBM traffic as secondary, clearly labelled fields.  achieved = wave-instructions per launch
(SQ_INSTS_VALU of the committed rocprofv3 counter pass, profiles/rNN_bench_pmc_summary.csv, stamped with the hash of the
sources it was taken from and marked stale when that is not the running library) / the kernel's duration measured LIVE in
this run with HIP events on the stream it runs on; peak = SIMDs x clock / 2 (one fp32 add / mul / fma wave-instruction per
2 cycles per SIMD, tools/bench_valu_issue.hip); frac <= 1.  `cpu_baseline` times the oracle (a CPU port of the loop — the
reference has no CPU renderer) on the host cores, on a bounded spp.
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


def is_bit_exact_variant(resolved_variant):
    """The streaming kernels (variants 2-5; 0 resolves to one of them) promise the oracle's bits.  Variant 1 (wave-per-pixel baseline: another
    summation order) and variant 6 (opt-in tolerance mode: products with the rounded reciprocal in the box tests) are held to north_star's
    tolerance |delta| < 1e-3 instead; whether their frame was bit-identical anyway is reported."""
    return resolved_variant in (2, 3, 4, 5)


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
                    3840, 2160, 10000, 50, "final scene of The Next Week (2401 quads, 1008 spheres, media, Perlin, image texture; not in the reference)",
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
    ap.add_argument("--sparse-parity", type=int, default=36,
                    help="N == 1: pixels of the last TIMED frame (full spp) checked against the CPU oracle after the timed region (0 = skip)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--pipeline", type=int, default=0, choices=[0, 1, 2],
                    help="frames in flight: 2 renders frame k+1 on a second HIP stream while frame k drains its last paths and is gathered "
                         "(two renderers, two shard buffers; the collectives stay on one stream, in order); 1 = strictly serial; "
                         "0 (default) = 1.  Measured on one GPU (round 3, non-blocking streams only): 72.8 instead of 74.7 ms per frame at N = 1 and "
                         "10.1 instead of 10.7 ms for a 1/8 shard (tools/shard_pipeline.py) — the next frame's ray generation fills the issue slots "
                         "and the tail of the draining frame.  Not the default: the per-kernel durations the roofline is priced with are those of "
                         "kernels that share the GPU (the line then carries serial_render_ms_rank0 too), and no multi-GPU run has exercised it")
    ap.add_argument("--verify-assembly", dest="verify_assembly", action="store_true", default=True,
                    help="N > 1 (default on): after the timed region rank 0 renders the frame alone and requires the assembled image to be the "
                         "same bits; the JSON line carries assembly_verified")
    ap.add_argument("--no-verify-assembly", dest="verify_assembly", action="store_false")
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


def profiled_counters(kernel_substr, workload, W, H, spp, depth, variant=0):
    """Per-launch hardware counters of the dominant kernel from the newest committed rocprofv3 summary of THIS workload at this
    frame size and depth (profiles/rNN_<name>_pmc_summary.csv, name = `bench` for the headline, else the workload; separate --pmc
    passes of this same command: tools/profile_round.sh, tools/profile_workload.sh).  A summary taken at another spp is scaled
    linearly to this run's samples (instruction counts are per sample; reported as counters_scaled_from_spp).
    Returns (counters dict, relative path, hash of the library build it was taken from, spp it was taken at) or four Nones."""
    import csv
    import glob
    legacy = {"bench": ("book1_final", 1200, 800, 500, 50), "book2_moving": ("book2_moving", 800, 800, 1000, 50),
              "book2_final": ("book2_final", 800, 800, 200, 40)}   # rounds 1-2: no config line (their one-pass profiles only)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.csv")), reverse=True):
        name = os.path.basename(path).split("_", 1)[1][:-len("_pmc_summary.csv")]
        lines = open(path).read().splitlines()
        stamp, cfg, cfg_variant = None, legacy.get(name), 0
        while lines and lines[0].startswith("#"):
            if "csrc_sha256:" in lines[0]:
                stamp = lines[0].split("csrc_sha256:")[1].strip()
            if "config:" in lines[0]:
                f = lines[0].split("config:")[1].split()
                cfg = (f[0], int(f[1]), int(f[2]), int(f[3]), int(f[4]))
                cfg_variant = int(f[5].split("=")[1]) if len(f) > 5 and f[5].startswith("variant=") else 0   # the kernel variant it was taken with
            lines.pop(0)
        if cfg is None or cfg[:3] != (workload, W, H) or cfg[4] != depth or cfg_variant != variant:
            continue
        # (profiles are taken at an spp that fits ONE pass, so mean_per_dispatch is the count of the whole cfg[3]-spp frame)
        v = {}
        for row in csv.DictReader(lines):
            if kernel_substr in row["kernel"]:
                v[row["counter"]] = float(row["mean_per_dispatch"]) * spp / cfg[3]
        if v:
            return v, os.path.relpath(path, ROOT), stamp, cfg[3]
    return None, None, None, None


def issue_roofline(v, kernel_ms, n_simd, clock_ghz):
    """The binding roof: vector-instruction issue.  achieved = SQ_INSTS_VALU / live kernel time; peak = one wave-instruction per
    2 cycles per SIMD at the device's peak engine clock (an upper bound of the clock the launch ran at, so frac is a lower
    bound of the issue utilisation at the actual clock)."""
    need = ("SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU")
    if v is None or any(k not in v for k in need):
        return None
    achieved = v["SQ_INSTS_VALU"] / (kernel_ms * 1e-3) / 1e9
    peak = n_simd * clock_ghz / 2.0
    lanes = v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_INSTS_VALU"] * 64.0)
    out = {"achieved": round(achieved, 2), "peak": round(peak, 2), "unit": "G wave-instructions/s", "frac": round(achieved / peak, 4),
           "wave_instructions_per_launch": v["SQ_INSTS_VALU"], "lanes_active_frac": round(lanes, 4),
           "lane_weighted_frac": round(achieved / peak * lanes, 4)}
    if "SQ_INSTS_SALU" in v:
        out["scalar_instructions_per_launch"] = v["SQ_INSTS_SALU"]
    if "SQ_LDS_IDX_ACTIVE" in v and "GRBM_GUI_ACTIVE" in v:
        cu_cycles = v["GRBM_GUI_ACTIVE"] / 8.0 * (n_simd / 4.0)   # GRBM_GUI_ACTIVE sums the 8 XCDs
        out["lds_busy_frac"] = round(v["SQ_LDS_IDX_ACTIVE"] / cu_cycles, 4)
        if "SQ_LDS_BANK_CONFLICT" in v:
            out["lds_bank_conflict_frac_of_busy"] = round(v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"], 4)
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


def cpu_leg(args, pkg, scene, cam, gpu_image_fn, bit_exact_contract=True):
    """cpu_baseline + algorithmic counts + parity sample, rank 0 at N == 1 only.  The oracle is the checker
    and the timed baseline here; it is never on the product path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    cores = host_cores()
    os.environ["ORC_PIN_THREADS"] = "1"   # one oracle thread per distinct CPU of the affinity mask (pinned, core count stated)
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
                      f"gcc -O2 -ffp-contract=off, {cores} pthreads over rows, each pinned to its own CPU"}
    parity = None
    if gpu_image_fn is not None:
        img = gpu_image_fn(spp)
        nan_mismatch = int(np.count_nonzero(np.isnan(img[..., :3]) != np.isnan(ref[..., :3])))   # NaN on one side only is a failure
        d = np.abs(img[..., :3] - ref[..., :3])
        both = ~np.isnan(d)
        parity = {"max_abs_delta": float(d[both].max()) if both.any() else float("nan"),
                  "mean_abs_delta": float(d[both].mean()) if both.any() else float("nan"),
                  "nan_mismatch": nan_mismatch, "nan_pixels_both": int(np.count_nonzero(np.isnan(img[..., :3]).any(-1) & np.isnan(ref[..., :3]).any(-1))),
                  "bit_identical": bool(nan_mismatch == 0 and np.all((img.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(img) & np.isnan(ref)))),
                  "tolerance": 1e-3, "sample": f"{W}x{H}x{spp}spp GPU vs CPU oracle, same seed"}
        if nan_mismatch or not (parity["max_abs_delta"] < 1e-3) or (bit_exact_contract and not parity["bit_identical"]):
            raise SystemExit(f"parity failed: {parity}")
    return base, counts, parity


def sparse_leg(args, frame_rgba, tolerance_only=False):
    """Parity of the TIMED frame itself (full spp): --sparse-parity pixels (the four corners + random ones) re-rendered by the
    CPU oracle at the full sample count and compared bit for bit.  The oracle is the checker, outside the timed region."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    W, H = args.width, args.height
    rng = np.random.default_rng(args.seed)
    gids = rng.integers(0, W * H, max(4, args.sparse_parity)).astype(np.uint32)
    gids[:4] = (0, W - 1, (H - 1) * W, H * W - 1)
    oscene, ocam = make_workload(args, O, oracle=True)
    t = time.perf_counter()
    exp = O.render_pixels(oscene.world, ocam, W, H, args.spp, args.depth, gids, args.seed)
    got = np.ascontiguousarray(frame_rgba.reshape(-1, 4)[gids])
    same = (got.view(np.uint32) == exp.view(np.uint32)) | (np.isnan(got) & np.isnan(exp))
    res = {"pixels": int(len(gids)), "spp": args.spp, "bit_identical": bool(same.all()), "mismatching_values": int((~same).sum()),
           "max_abs_delta": float(np.nanmax(np.abs(got - exp))), "oracle_seconds": round(time.perf_counter() - t, 2),
           "sample": "pixels of the last timed frame vs O.render_pixels at the full sample count, same seed"}
    # the streaming kernels' contract (variants 2-5; 0 resolves to one of them) is the oracle's bits; the wave-per-pixel baseline (variant 1:
    # another summation order) and the opt-in tolerance mode (variant 6) are held to the 1e-3 tolerance (is_bit_exact_variant above)
    if not res["bit_identical"] and (not tolerance_only or not (res["max_abs_delta"] < 1e-3)):
        raise SystemExit(f"full-spp sparse parity failed: {res}")
    return res


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: start `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py
    <same arguments>` as a CHILD process — one rank per GPU — relay its output and return its exit code.  Nothing in THIS process has touched
    the GPU at this point (no torch import, no HIP call, the product library is not loaded), and it never does: a process that has
    initialised the GPU must not be replaced or re-executed on this pool."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    print(f"bench.py: no launcher in the environment, starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env, cwd=ROOT).returncode


def main():
    args = parse()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL on this pool; must precede any HIP call
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(self_launch(args))   # before `import torch` and before the product library is loaded: this process stays off the GPU
    import torch
    import torch.distributed as dist

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world_size:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world_size} of the launcher")
    if args.same_device:
        local_rank = 0
    n_visible = torch.cuda.device_count()   # (counting devices does not initialise the GPU on this image)
    if n_visible and local_rank >= n_visible:
        # a launcher that isolates the ranks (one visible device per process: HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES set per rank) hands
        # every rank "device 0" of its own view; with fewer visible devices than ranks for any other reason RCCL itself refuses later
        local_rank %= n_visible
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
    # With one frame in flight everything runs on torch's current stream.  With two, each slot renders on its own non-blocking stream and —
    # for N > 1 — the collectives and the assembly stay on ONE further non-blocking stream, in frame order.  (The legacy default stream is
    # kept out of it: event waits / records on it were measured to serialise the two side streams.)
    stream = torch.cuda.current_stream() if depth == 1 else torch.cuda.Stream(device=dev)
    side = [torch.cuda.Stream(device=dev) for _ in range(depth)] if depth > 1 else [stream]
    images = [torch.zeros(H * W * 4, dtype=torch.float32, device=dev) if rank == 0 else None for _ in range(depth)]
    image = images[0]
    from ray_tracing_v06_amd import multigpu
    shards = [torch.zeros(r.shard_floats(), dtype=torch.float32, device=dev) for _ in range(depth)] if world_size > 1 else None
    rendered = [torch.cuda.Event() for _ in range(depth)]
    consumed = [torch.cuda.Event() for _ in range(depth)]
    torch.cuda.synchronize()   # the buffers above were zeroed on the current stream
    for e in consumed:
        e.record(stream)

    kernel_events = []
    exchange_events = []   # N > 1: (render done on this rank, gather [+ assembly on rank 0] done) on the collectives' stream
    frame = [0]

    def step(record):
        d = frame[0] % depth
        frame[0] += 1
        sd = side[d]
        if world_size > 1:
            sd.wait_event(consumed[d])        # the buffers of this slot are free again (frame k - depth was gathered / assembled)
        e0 = e1 = None
        if record:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(sd)
        target = images[d] if world_size == 1 else shards[d]
        rs[d].render_async(sd.cuda_stream, target.data_ptr())
        if record:
            e1.record(sd)
            kernel_events.append((e0, e1))
        if world_size > 1:                    # N == 1: a slot's renderer and image are touched by its own stream only, in order
            rendered[d].record(sd)
            stream.wait_event(rendered[d])    # everything after the render stays on ONE stream, in frame order
            with torch.cuda.stream(stream):
                if record:
                    g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    g0.record(stream)
                gathered = multigpu.gather_shards(shards[d], world_size, rank, dst=0)  # the single frame-end exchange (RCCL over xGMI)
                if rank == 0:
                    rs[d].assemble(gathered.data_ptr(), images[d].data_ptr(), stream.cuda_stream)
                if record:
                    g1.record(stream)
                    exchange_events.append((g0, g1))
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
    serial_ms = None
    if depth > 1:   # for the record: the same frame strictly one at a time (rank-local render, no gather), after the timed region
        fence()
        ts = time.perf_counter()
        for _ in range(3):
            rs[0].render_async(side[0].cuda_stream, (images[0] if world_size == 1 else shards[0]).data_ptr())
            side[0].synchronize()
        serial_ms = (time.perf_counter() - ts) / 3 * 1e3
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in kernel_events])) if kernel_events else float("nan")
    kernel_ms_source = "HIP events around every launch of the timed region"
    # per-kernel durations of the timed steps: HIP events the renderer records on the stream its kernels run on (ring of 32 renders)
    per_kernel = []
    kinfo = rs[0].kernel_info()
    pinfo = rs[0].pass_info()
    if kinfo["variant"] >= 2:   # the resolved variant: a world the streaming kernels cannot take falls back to the baseline kernel under variant 0
        n_back = min(args.steps, 32 * depth)
        for k in range(n_back):
            rr = rs[(frame[0] - 1 - k) % depth]
            per_kernel.append(rr.kernel_times(k // depth))
    primary_ms, stream_ms, resolve_ms = (float(np.mean([t[i] for t in per_kernel])) for i in range(3)) if per_kernel else (float("nan"),) * 3
    per_rank = None
    if world_size > 1:
        # every rank's own numbers, so that a scaling curve can be read: render (all kernels of a step, HIP events on the render's stream), the three
        # kernels, and the frame-end exchange as this rank sees it (from "my render is done" to "gather [+ assembly on rank 0] done": on rank 0 it
        # contains the wait for the slowest peer)
        exch_ms = float(np.mean([a.elapsed_time(b) for a, b in exchange_events])) if exchange_events else float("nan")
        mine = torch.tensor([kernel_ms, primary_ms, stream_ms, resolve_ms, exch_ms, float(local_rank), float(torch.cuda.current_device())],
                            dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        every = [torch.zeros_like(mine) for _ in range(world_size)]
        dist.all_gather(every, mine)
        per_rank = np.stack([e.cpu().numpy() for e in every])
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
            base, counts, parity = cpu_leg(args, pkg, scene, cam, None if args.no_parity else gpu_image, bit_exact_contract=is_bit_exact_variant(kinfo["variant"]))
        value = total_samples * args.steps / elapsed / 1e6
        out = {
            "metric": "Msamples/sec (WxHxspp) on Book-1 final scene" if args.workload == "book1_final" else f"Msamples/sec (WxHxspp) on {args.workload}",
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{WORKLOADS[args.workload][8]}, {W}x{H}, {spp} spp, "
                                   f"max_depth {args.depth}, seed {args.seed}, {WORKLOADS[args.workload][9]}",
                       "parallelism": (f"tile-shard x{world_size} + 1 {'RCCL' if args.backend == 'nccl' else args.backend} gather" if world_size > 1 else "single GPU"),
                       "frames_in_flight": depth, "serial_render_ms_rank0": (None if serial_ms is None else round(serial_ms, 3)),
                       "kernel_variant": args.variant},
            "kernel_ms_per_step_rank0": round(kernel_ms, 3), "kernel_ms_source": kernel_ms_source + " (all kernels of a step)",
        }
        if counts is None:
            counts = count_leg(args)
        if counts is not None:
            bytes_per_sample = 32.0 * counts["box_tests"] + 16.0 * counts["leaf_tests"] + 16.0 * counts["shaded_hits"] + 16.0 / spp
            launch_samples = total_samples / world_size
            dominant = "render_kernel_xchg" if kinfo["variant"] == 5 else "render_kernel_stream"
            # counters of the kernel variant that ran (a summary names the variant it was taken with; none = the default, 0)
            pmc, pmc_src, pmc_stamp, pmc_spp = profiled_counters(dominant, args.workload, W, H, spp, args.depth, args.variant)
            if pmc is not None and world_size > 1:
                # a rank's launch traces 1/N of the frame's samples (tiles interleave finely, so a shard is a fair sample of the frame): the
                # committed whole-frame counters are scaled to the launch like samples_per_launch
                pmc = {k: v / world_size for k, v in pmc.items()}
            info = pkg.api.device_info(local_rank)
            n_simd, clock_ghz = info["compute_units"] * 4, info["clock_khz"] / 1e6
            roof = issue_roofline(pmc, stream_ms, n_simd, clock_ghz) or {"achieved": None, "peak": round(n_simd * clock_ghz / 2.0, 2),
                                                                          "unit": "G wave-instructions/s", "frac": None}
            primary_bytes = 48                                        # PRIMARY_BYTES (csrc/rt_device.hip): origin|time, direction, RNG state
            slot_bytes = int(pinfo["bytes_per_sample"]) - primary_bytes   # RT_SAMPLE_BYTES: 12 (one global_store_dwordx3 per finished sample)
            traffic = traffic_true_reads = None
            if pmc and "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
                # MI355X_MICROARCH.md, HBM section: both counters are in KB; on gfx950 FETCH_SIZE counts 64 B per 128-B request on wide reads
                traffic = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
                # this kernel's reads are the primary-ray records only (48 B per sample, non-temporal 16-B loads, each byte read once; the scene
                # image is 60 KB per workgroup): for THOSE loads the x2 correction over-states the traffic (raw FETCH_SIZE = 0.65 of the true bytes)
                traffic_true_reads = primary_bytes * launch_samples + pmc["WRITE_SIZE"] * 1024.0
            lib_hash = pkg.capi.library_hash()   # embedded in the loaded librt06.so at build time: the BINARY's provenance
            out["roofline"] = {
                "bound": "valu_issue", "kernel": dominant, **roof, "traffic": traffic,
                "traffic_with_true_read_bytes": traffic_true_reads,
                "traffic_note": "traffic = (2 x FETCH_SIZE + WRITE_SIZE) KB as the guide prescribes; traffic_with_true_read_bytes = 48 B x samples actually "
                                "read (the primary-ray records, each once) + WRITE_SIZE: the guide's x2 FETCH correction over-states non-temporal 16-B loads",
                "kernel_ms": round(stream_ms, 3), "kernel_ms_source": "HIP events on the kernel's stream, summed over the passes of a step, mean over the timed steps (rt_renderer_kernel_times)",
                "passes_per_step": pinfo["n_passes"], "spp_per_pass": pinfo["pass_spp"],
                "counters_scaled_from_spp": (None if pmc_spp in (None, spp) else pmc_spp),
                "other_kernels_ms": {"primary_rays_kernel": round(primary_ms, 3), "resolve_kernel": round(resolve_ms, 3)},
                # the two short kernels of a step are HBM streams: algorithmic bytes per sample (the primary-ray record written, the sample slot
                # read — both from rt_renderer_pass_info, i.e. from the constants the kernels are compiled with) over their live durations
                "other_kernels_hbm": {
                    "primary_rays_kernel": {"algorithmic_bytes_per_sample": primary_bytes, "GBps": round(primary_bytes * launch_samples / (primary_ms * 1e-3) / 1e9, 1),
                                            "frac_of_peak": round(primary_bytes * launch_samples / (primary_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                    "resolve_kernel": {"algorithmic_bytes_per_sample": slot_bytes, "GBps": round(slot_bytes * launch_samples / (resolve_ms * 1e-3) / 1e9, 1),
                                       "frac_of_peak": round(slot_bytes * launch_samples / (resolve_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}},
                "peak_definition": f"{n_simd} SIMDs x {clock_ghz:.3f} GHz / 2 cycles per wave64 fp32 add/mul/fma (tools/bench_valu_issue.hip); compares, selects, min/max "
                                   "issue in 4 cycles and scalar instructions are not hidden, so frac = 1 is not reachable by this instruction mix; measured: a pure "
                                   "v_fma_f32 stream (128 per loop iteration, tools/bench_valu_peak.hip, profiles/r03_valu_peak_microbench.txt) sustains 898 G/s "
                                   "= 0.73 of this peak at the kernel's occupancy (6 waves per SIMD; 950 G/s at 8), because the shader clock sags to ~1.95 GHz under it",
                "samples_per_launch": launch_samples,
                "counters_source": pmc_src, "counters_source_csrc_sha256": pmc_stamp, "library_csrc_sha256": lib_hash,
                "library_matches_tree_sources": bool(lib_hash == pkg.capi.source_hash()),
                "counters_stale": (None if pmc_src is None else bool(pmc_stamp != lib_hash)),
                "counts_per_sample": {k: round(v, 3) for k, v in counts.items()},
                "hbm": {"note": ("secondary: the scene is LDS-resident, the algorithmic bytes are served from the LDS and HBM is not the roof" if kinfo["lds_resident"] else
                                 "secondary: the scene's records are read through the vector L1 / L2 (global-memory form of the kernel, top of the tree in the LDS); the algorithmic bytes are cache-served and HBM is not the roof"),
                        "algorithmic_bytes_per_sample": round(bytes_per_sample, 1),
                        "algorithmic_GBps": round(bytes_per_sample * launch_samples / (stream_ms * 1e-3) / 1e9, 1),
                        "measured_bytes_per_launch": traffic,
                        "measured_GBps": None if traffic is None else round(traffic / (stream_ms * 1e-3) / 1e9, 1),
                        "peak_GBps": HBM_PEAK_GBS,
                        "measured_frac_of_peak": None if traffic is None else round(traffic / (stream_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
            if per_rank is not None:
                # N > 1: what a scaling curve needs to be read.  Row r = rank r's own measurement.
                rms = per_rank[:, 0]
                out["ranks_seen"] = int(dist.get_world_size())
                out["backend"] = str(dist.get_backend())
                out["per_rank"] = {
                    "render_ms": [round(float(x), 3) for x in rms],
                    "render_ms_min": round(float(rms.min()), 3), "render_ms_max": round(float(rms.max()), 3),
                    "render_imbalance": round(float(rms.max() / rms.min() - 1.0), 4),
                    "primary_rays_kernel_ms": [round(float(x), 3) for x in per_rank[:, 1]],
                    "dominant_kernel_ms": [round(float(x), 3) for x in per_rank[:, 2]],
                    "resolve_kernel_ms": [round(float(x), 3) for x in per_rank[:, 3]],
                    "exchange_ms": [round(float(x), 3) for x in per_rank[:, 4]],
                    "device": [int(x) for x in per_rank[:, 6]],
                    "note": "HIP events on each rank's own streams, mean over the timed steps; exchange_ms = from 'this rank's render is done' to 'the "
                            "gather (+ assemble_kernel on rank 0) is done' on the collectives' stream: rank 0's contains the wait for the slowest peer"}
                out["gather_assemble_ms_rank0"] = round(float(per_rank[0, 4]), 3)
                out["step_ms_breakdown_rank0"] = {"render": round(float(per_rank[0, 0]), 3), "exchange_and_assemble": round(float(per_rank[0, 4]), 3),
                                                  "host_and_launch_gaps": round(elapsed / args.steps * 1e3 - float(per_rank[0, 0]) - float(per_rank[0, 4]), 3)}
                out["roofline"]["kernel_ms_per_rank"] = [round(float(x), 3) for x in per_rank[:, 2]]
                out["roofline"]["counters_scaled_by"] = f"1/{world_size} of the whole-frame counter summary (a rank's launch traces 1/{world_size} of the samples)"
            if base is not None:
                out["cpu_baseline"] = base
            if parity is not None:
                out["parity"] = parity
            if world_size == 1 and args.sparse_parity > 0 and not args.no_parity and args.steps > 0:
                out["parity_timed_frame"] = sparse_leg(args, images[(frame[0] - 1) % depth].cpu().numpy(), tolerance_only=not is_bit_exact_variant(kinfo["variant"]))
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
        dist.barrier()   # rank 0 may still be verifying the assembled frame: nobody tears the group down before it is done
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

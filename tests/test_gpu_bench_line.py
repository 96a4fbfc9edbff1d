"""bench.py end to end on the GPU at a small size: the ONE JSON line carries what the contract asks for (metric, value, roofline, cpu_baseline,
parity of a full frame and of the timed frame at its full spp), serial and with two frames in flight, and for a multi-pass workload."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(*args, env=None):
    e = dict(os.environ, **(env or {}))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("pipeline", ["1", "2"])
def test_headline_line_at_a_small_size(pipeline):
    d = _bench("--width", "240", "--height", "160", "--spp", "24", "--steps", "4", "--warmup", "1", "--cpu-seconds", "1", "--pipeline", pipeline)
    assert d["metric"].startswith("Msamples/sec") and d["unit"] == "Msamples/s" and d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "strong" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert abs(d["value"] - 240 * 160 * 24 / d["ms_per_step"] / 1e3) < 1e-2 * d["value"]
    assert d["config"]["frames_in_flight"] == int(pipeline) and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "valu_issue" and r["kernel"] == "render_kernel_stream" and r["peak"] > 1000 and r["passes_per_step"] == 1
    assert r["kernel_ms"] > 0 and set(r["other_kernels_ms"]) == {"primary_rays_kernel", "resolve_kernel"}
    assert len(r["library_csrc_sha256"]) == 64 and r["library_matches_tree_sources"] is True
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and "spp" in c["sample"]
    assert d["parity"]["bit_identical"] is True and d["parity"]["nan_mismatch"] == 0 and d["parity"]["max_abs_delta"] == 0.0
    t = d["parity_timed_frame"]
    assert t["bit_identical"] is True and t["spp"] == 24 and t["pixels"] == 36
    if pipeline == "2":
        assert d["config"]["serial_render_ms_rank0"] > 0


def _fracs(node, path=""):
    """every (path, value) of the line whose key is a roofline fraction"""
    if isinstance(node, dict):
        for k, v in node.items():
            if k in ("frac", "frac_of_peak", "lane_weighted_frac", "lanes_active_frac", "measured_frac_of_peak", "lds_busy_frac") and v is not None:
                yield path + "/" + k, v
            else:
                yield from _fracs(v, path + "/" + k)


def test_every_roofline_fraction_of_the_headline_line_is_a_fraction():
    """the headline geometry (1200x800, depth 50) at a reduced spp: the committed counter summary applies (scaled by spp), and no `frac` /
    `frac_of_peak` of the line — dominant kernel, the two HBM-stream side kernels, the measured HBM traffic — may leave (0, 1]; the sample slot the
    resolve kernel is priced with is the 12 bytes the kernels are compiled with"""
    d = _bench("--spp", "40", "--steps", "3", "--warmup", "1", "--cpu-seconds", "1", "--sparse-parity", "8")
    r = d["roofline"]
    fr = dict(_fracs(d))
    assert "/roofline/frac" in fr and "/roofline/other_kernels_hbm/resolve_kernel/frac_of_peak" in fr, fr
    for k, v in fr.items():
        assert 0.0 < v <= 1.0, (k, v)
    assert r["other_kernels_hbm"]["resolve_kernel"]["algorithmic_bytes_per_sample"] == 12
    assert r["other_kernels_hbm"]["primary_rays_kernel"]["algorithmic_bytes_per_sample"] == 48
    assert r["counters_scaled_from_spp"] == 500 and r["traffic"] is not None
    # the true read bytes (48 B x samples, each read once) + the counted writes, next to the guide's (2 x FETCH + WRITE)
    assert r["traffic_with_true_read_bytes"] >= 48 * 1200 * 800 * 40 and r["traffic_with_true_read_bytes"] < r["traffic"]
    assert d["parity"]["bit_identical"] is True and d["parity_timed_frame"]["bit_identical"] is True


def test_tolerance_mode_line_names_its_variant_and_its_own_counters():
    """`bench.py --variant 6` (opt-in tolerance mode): the line says which kernel variant ran, prices the roofline with the counter summary taken WITH that
    variant (profiles/rNN_bench_v6_pmc_summary.csv, `# config: ... variant=6`), and its parity legs are held to 1e-3 and report bit-identity"""
    d = _bench("--variant", "6", "--spp", "40", "--steps", "3", "--warmup", "1", "--cpu-seconds", "1", "--sparse-parity", "8")
    assert d["config"]["kernel_variant"] == 6
    r = d["roofline"]
    assert r["counters_source"].endswith("bench_v6_pmc_summary.csv") and 0.0 < r["frac"] <= 1.0
    assert d["parity"]["max_abs_delta"] < 1e-3 and d["parity"]["tolerance"] == 1e-3 and isinstance(d["parity"]["bit_identical"], bool)
    assert d["parity_timed_frame"]["max_abs_delta"] < 1e-3
    e = _bench("--spp", "40", "--steps", "3", "--warmup", "1", "--cpu-seconds", "0", "--no-parity")
    assert e["config"]["kernel_variant"] == 0 and e["roofline"]["counters_source"].endswith("bench_pmc_summary.csv")
    # fewer vector instructions per launch than the default kernel's summary: the products replace the near quotients too
    assert r["wave_instructions_per_launch"] < e["roofline"]["wave_instructions_per_launch"]


def test_multi_pass_workload_line(monkeypatch):
    """three passes per step: the per-kernel times are sums over the passes and the counters are found by workload + size + depth"""
    d = _bench("--workload", "cornell_box", "--spp", "30", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1", env={"RT06_PASS_SPP": "10"})
    r = d["roofline"]
    assert r["passes_per_step"] == 3 and r["spp_per_pass"] == 10
    assert sum(r["other_kernels_ms"].values()) + r["kernel_ms"] <= 1.05 * d["kernel_ms_per_step_rank0"]
    assert r["counters_source"].endswith("cornell_box_pmc_summary.csv") and r["counters_scaled_from_spp"] == 1000 and 0.3 < r["frac"] <= 1.0
    assert d["parity"]["bit_identical"] is True and d["parity_timed_frame"]["bit_identical"] is True


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("ranks,pipeline", [(2, "1"), (3, "1"), (2, "2")])
def test_n_gt_1_line_rehearsed_on_one_gpu(ranks, pipeline):
    """bench.py's N > 1 path as the driver launches it (torch.distributed.run, one process per rank), rehearsed on this one-GPU box: every rank
    renders its tile shard on cuda:0, the frame-end gather goes over gloo instead of RCCL (RCCL refuses two ranks on one device), rank 0 assembles
    and — after the timed region — verifies the assembled frame against the frame it renders alone, bit for bit.  Everything of the N > 1 path
    but the transport itself: shard buffers, stream / event ordering (also with two frames in flight), max-over-ranks timing, the ONE JSON line."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--backend", "gloo", "--same-device",
           "--width", "243", "--height", "161", "--spp", "12", "--steps", "3", "--warmup", "1", "--cpu-seconds", "0", "--pipeline", pipeline]
    out = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["steps"] == 3
    _check_n_gt_1_line(d, ranks, 243 * 161 * 12, pipeline)


def _check_n_gt_1_line(d, ranks, samples, pipeline="1"):
    assert d["n_gpus"] == ranks and d["ranks_seen"] == ranks and d["backend"] == "gloo" and d["scaling"] == "strong" and d["value"] > 0
    assert abs(d["value"] - samples / d["ms_per_step"] / 1e3) < 1e-2 * d["value"]
    assert d["assembly_verified"] is True and d["config"]["frames_in_flight"] == int(pipeline)
    pr = d["per_rank"]
    for k in ("render_ms", "dominant_kernel_ms", "primary_rays_kernel_ms", "resolve_kernel_ms", "exchange_ms", "device"):
        assert len(pr[k]) == ranks, (k, pr[k])
    assert 0 < pr["render_ms_min"] <= pr["render_ms_max"] and pr["render_imbalance"] >= 0 and d["gather_assemble_ms_rank0"] > 0
    r = d["roofline"]
    assert len(r["kernel_ms_per_rank"]) == ranks and all(x > 0 for x in r["kernel_ms_per_rank"])
    assert abs(r["samples_per_launch"] * ranks - samples) < 1e-6 * samples
    for k, v in _fracs(d):
        assert 0.0 < v <= 1.0, (k, v)


def test_bare_invocation_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher in the environment (the way the driver calls the N = 1 case): bench.py starts
    torch.distributed.run as a child before it touches the GPU and relays the one JSON line and the exit code."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device", "--width", "243",
                          "--height", "161", "--spp", "12", "--steps", "3", "--warmup", "1"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    assert "starting 2 ranks" in out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    _check_n_gt_1_line(json.loads(lines[0]), 2, 243 * 161 * 12)


# The shapes of the driver's scaling run (SCALE: N = 1, 2, 4, 8 on the headline frame; BASELINE configs[3] on 4 and configs[4] on 8 GPUs), rehearsed
# through bench.py itself on this one-GPU box: one process per rank, every rank on cuda:0, gloo as the transport.  N = 8 cannot be rehearsed this
# way: the pool's process guard admits 6 processes on the card and the test runner is one of them (8 ranks on one device are covered behind the
# C ABI with the memcpy transport instead: tests/test_multi_gpu_c.py).
@pytest.mark.parametrize("workload,ranks,size,spp,pipeline,env", [
    ("book1_final", 4, (1200, 800), 16, "1", {}),
    ("book1_final", 4, (1200, 800), 16, "2", {}),
    ("cornell_box", 4, (600, 600), 24, "1", {}),                       # configs[3]: 600x600 tile-sharded over 4 ranks
    ("book2_final", 4, (3840, 2160), 6, "1", {"RT06_PASS_SPP": "2"}),  # configs[4]'s frame: 3 passes per rank, running sums, 16.6 MB x 2 per peer
])
def test_scale_shapes_rehearsed_through_bench(workload, ranks, size, spp, pipeline, env):
    args = ["--gpus", str(ranks), "--backend", "gloo", "--same-device", "--workload", workload, "--width", str(size[0]), "--height", str(size[1]),
            "--spp", str(spp), "--steps", "2", "--warmup", "1", "--pipeline", pipeline]
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, env=dict(e, **env), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    _check_n_gt_1_line(d, ranks, size[0] * size[1] * spp, pipeline)
    if env:
        assert d["roofline"]["passes_per_step"] == 3 and d["roofline"]["spp_per_pass"] == 2

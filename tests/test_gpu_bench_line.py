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
    assert d["n_gpus"] == ranks and d["steps"] == 3 and d["scaling"] == "strong" and d["value"] > 0
    assert abs(d["value"] - 243 * 161 * 12 / d["ms_per_step"] / 1e3) < 1e-2 * d["value"]
    assert d["assembly_verified"] is True
    assert d["config"]["frames_in_flight"] == int(pipeline)

"""N > 1 path on CPU: world_size-2 (and 3) gloo processes run the product's frame-end gather
(ray-tracing-v06_amd/multigpu.py) on tile shards and must reproduce the full frame bit for bit.
The render kernels themselves are covered on the GPU by test_tile_sharding_is_gpu_count_invariant."""
import os
import socket
import subprocess
import sys

import pytest

from _common import ROOT, pkg


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world_size", [2, 3])
def test_gather_of_tile_shards_reassembles_the_frame(world_size):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world_size}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "tests", "_dist_worker.py")]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]


def test_tile_layout_matches_the_library():
    p = pkg()
    from ray_tracing_v06_amd import multigpu
    assert multigpu.tile_layout(1200, 800, 8) == (150, 15000, 1875, 1875 * 256)
    assert multigpu.tile_layout(203, 117, 3) == (26, 390, 130, 130 * 256)


def test_bench_without_a_launcher_starts_its_ranks_as_a_child_process():
    """`python bench.py --gpus 2` with no launcher in the environment must not die in argument handling: it starts torch.distributed.run as a
    child (before importing torch or loading the HIP library) and returns the child's exit code.  Here there is no GPU, so the ranks themselves
    fail — loudly, with a non-zero code that bench.py relays; the GPU twin (tests/test_gpu_bench_line.py) checks the JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device", "--width", "32",
                          "--height", "16", "--spp", "1", "--steps", "1", "--warmup", "0"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert "starting 2 ranks" in res.stderr and "torch.distributed.run" in res.stderr
    import torch
    if not torch.cuda.is_available():
        assert res.returncode != 0 and not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]


def test_gather_of_tile_shards_with_eight_ranks():
    """the rank count of the scaling run's last point (N = 8), on the CPU: eight gloo processes, one gather, the frame back bit for bit"""
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "tests", "_dist_worker.py")]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]

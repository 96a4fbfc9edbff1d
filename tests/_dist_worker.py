"""Worker of tests/test_dist_gloo.py: one of WORLD_SIZE CPU processes (backend gloo).

Each rank cuts ITS tiles out of a full reference frame (the CPU oracle's render — the oracle is test
infrastructure and this is a test), runs the product's frame-end gather, and rank 0 re-assembles the frame
with an independent numpy index computation and requires the original bits back."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as G  # noqa: E402
import _oracle as O  # noqa: E402


def shard_of(image, rank, world_size, tiles_x, n_tiles, n_local):
    H, W, _ = image.shape
    shard = np.zeros((n_local, 64, 4), np.float32)
    for tl in range(n_local):
        gt = tl * world_size + rank
        if gt >= n_tiles:
            continue
        tx, ty = gt % tiles_x, gt // tiles_x
        for p in range(64):
            x, y = tx * 8 + p % 8, ty * 8 + p // 8
            if x < W and y < H:
                shard[tl, p] = image[y, x]
    return shard.reshape(-1)


def assemble(gathered, W, H, world_size, tiles_x, n_local):
    shards = gathered.reshape(world_size, n_local, 64, 4)
    out = np.zeros((H, W, 4), np.float32)
    for y in range(H):
        for x in range(W):
            gt = (y // 8) * tiles_x + x // 8
            out[y, x] = shards[gt % world_size, gt // world_size, (y % 8) * 8 + x % 8]
    return out


def main():
    dist.init_process_group(backend="gloo")
    rank, world_size = dist.get_rank(), dist.get_world_size()
    pkg = G.load_package()
    from ray_tracing_v06_amd import multigpu
    W, H, spp = 77, 45, 3  # ragged: 10 x 6 tiles
    scene = O.Scene.book1_final(1984)
    cam = O.camera_defocus((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, W / H, 0.1, 10.0)
    full, _ = O.render(scene.world, cam, W, H, spp, 50, threads=2)
    tiles_x, n_tiles, n_local, shard_floats = multigpu.tile_layout(W, H, world_size)
    shard = torch.from_numpy(shard_of(full, rank, world_size, tiles_x, n_tiles, n_local))
    assert shard.numel() == shard_floats
    gathered = multigpu.gather_shards(shard, world_size, rank, dst=0)
    ok = 1
    if rank == 0:
        got = assemble(gathered.numpy(), W, H, world_size, tiles_x, n_local)
        ok = int(got.tobytes() == full.tobytes())
    else:
        assert gathered is None
    flag = torch.tensor([ok])
    dist.broadcast(flag, src=0)
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()

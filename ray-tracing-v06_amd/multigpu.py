"""Frame-end exchange of the tile-sharded renderer: one process per GPU, one RCCL gather over xGMI.

The reference has no multi-GPU path.  Here the frame is cut into 8x8-pixel tiles; tile t
(row-major tile order) belongs to rank t % world_size, so sky-heavy and ground-heavy regions interleave
finely across ranks.  Every rank holds the whole flat scene (tens of KB), renders its tiles into a compact
tile-major shard of identical size on every rank, and the only communication is `gather_shards`: a single
gather of those shards to rank 0, followed by `Renderer.assemble` (a de-interleave kernel) on rank 0.
The counter-based RNG is keyed by the GLOBAL pixel id and sample index, so the assembled image has the same
bits for any number of GPUs.  torch.distributed is plumbing only: backend "nccl" is RCCL on ROCm; the same
code runs on "gloo" CPU tensors in the tests.
"""
import torch
import torch.distributed as dist

TILE = 8  # RT_TILE in csrc/rt_internal.hpp


def tile_layout(width, height, world_size):
    """(tiles_x, n_tiles, n_local_tiles, shard_floats) — the host-side view of TileMap (csrc/rt_render_kernels.hpp)."""
    tiles_x = (width + TILE - 1) // TILE
    tiles_y = (height + TILE - 1) // TILE
    n_tiles = tiles_x * tiles_y
    n_local = (n_tiles + world_size - 1) // world_size
    return tiles_x, n_tiles, n_local, n_local * TILE * TILE * 4


def gather_shards(shard, world_size, rank, dst=0, group=None):
    """The single frame-end collective.  `shard`: this rank's flat float32 shard (same length on every rank).
    Returns the rank-major concatenation of all shards on `dst`, None elsewhere."""
    if world_size == 1:
        return shard
    if shard.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal path (bench.py --backend gloo --same-device on a one-GPU box): gloo gathers host tensors
        out = gather_shards(shard.cpu(), world_size, rank, dst, group)
        return out.to(shard.device) if out is not None else None
    if rank == dst:
        gathered = torch.empty(shard.numel() * world_size, dtype=shard.dtype, device=shard.device)
        dist.gather(shard, list(gathered.chunk(world_size)), dst=dst, group=group)
        return gathered
    dist.gather(shard, None, dst=dst, group=group)
    return None

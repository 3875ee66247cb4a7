"""Multi-GPU sharding of a slide's tiles: one process per GPU, contiguous tile shards,
no data-path collective inside the encoders, ONE all-gather of the per-tile embeddings for the
slide-level steps that need every tile (prompt screening means, spatial refine).

Works with any initialised ``torch.distributed`` backend: ``nccl`` (= RCCL over xGMI on MI355X) for
GPU tensors, ``gloo`` for the CPU tests.  The reference has no inference-side multi-GPU code
(WSI scripts pin ``device='cuda:0'``, e.g. zeroshot_subtyping_WSI.py:26); this is new.
"""
from __future__ import annotations

from typing import Callable, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition; the first ``n_items % world`` ranks get one extra item."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"rank {rank} / world {world}")
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_capacity(n_items: int, world: int) -> int:
    return -(-n_items // world)


def all_gather_rows(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """Gather row-sharded ``local`` ([n_local, D], partition = shard_bounds) into [n_total, D] on every rank.

    Shards are padded to the common capacity so a single ``all_gather_into_tensor`` moves everything.
    """
    if not (dist.is_available() and dist.is_initialized()):
        if local.shape[0] != n_total:
            raise ValueError("not distributed: local must hold every row")
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(n_total, rank, world)
    if local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} holds {local.shape[0]} rows, expected {hi - lo}")
    cap = shard_capacity(n_total, world)
    D = local.shape[1]
    send = local.new_zeros((cap, D))
    send[: hi - lo] = local
    recv = local.new_empty((world * cap, D))
    dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
    out = local.new_empty((n_total, D))
    for r in range(world):
        a, b = shard_bounds(n_total, r, world)
        out[a:b] = recv[r * cap: r * cap + (b - a)]
    return out


def encode_tiles_sharded(encode: Callable[[torch.Tensor], torch.Tensor], n_tiles: int,
                         load_tiles: Callable[[int, int], torch.Tensor], batch: int = 256, group=None) -> torch.Tensor:
    """Each rank encodes tiles [lo, hi) in batches with ``encode`` (e.g. KEEPModel.encode_image) and all
    ranks receive the full [n_tiles, D] embedding matrix."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    lo, hi = shard_bounds(n_tiles, rank, world)
    parts = [encode(load_tiles(s, min(s + batch, hi))) for s in range(lo, hi, batch)]
    if parts:
        local = torch.cat(parts, dim=0)
    else:
        probe = encode(load_tiles(0, 1))            # only to learn D / dtype / device for an empty shard
        local = probe[:0]
    return all_gather_rows(local, n_tiles, group)

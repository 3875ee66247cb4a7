"""Shard / all-gather logic of the multi-GPU path on the gloo backend (world_size 2 and 3, CPU)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from keep_amd.distributed import all_gather_rows, encode_tiles_sharded, shard_bounds, shard_capacity


def test_shard_bounds_cover_everything_once():
    for n in (0, 1, 5, 197, 1000, 100_000):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) == shard_capacity(n, world) or n == 0
    with pytest.raises(ValueError):
        shard_bounds(10, 3, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def fake_encode(tiles: torch.Tensor) -> torch.Tensor:
    """Stand-in for encode_image: a deterministic function of the tile content only."""
    flat = tiles.reshape(tiles.shape[0], 48).double()
    return torch.stack([flat.sum(1), (flat * flat).sum(1), flat[:, 0], flat[:, -1]], dim=1).float()


def _worker(rank, world, port, n_tiles, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(7)
        all_tiles = torch.randn(n_tiles, 3, 4, 4, generator=g)
        out = encode_tiles_sharded(fake_encode, n_tiles, lambda a, b: all_tiles[a:b], batch=16)
        ok = torch.equal(out, fake_encode(all_tiles))
        lo, hi = shard_bounds(n_tiles, rank, world)
        gathered = all_gather_rows(fake_encode(all_tiles[lo:hi]), n_tiles)
        ok = ok and torch.equal(gathered, fake_encode(all_tiles))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_tiles", [(2, 101), (2, 64), (3, 50), (2, 1), (3, 2), (2, 33)])
def test_sharded_encode_matches_single_process(world, n_tiles):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_tiles, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    results = dict(q.get(timeout=5) for _ in range(world))
    assert all(results[r] for r in range(world)), results


def test_owner_offset_map_is_the_inverse_of_shard_bounds():
    from keep_amd.distributed import _owner_and_offset
    for n in (1, 2, 7, 64, 101, 1000):
        for world in (1, 2, 3, 8):
            owner, off = _owner_and_offset(n, world, "cpu")
            for r in range(world):
                lo, hi = shard_bounds(n, r, world)
                assert torch.equal(owner[lo:hi], torch.full((hi - lo,), r)) and torch.equal(off[lo:hi], torch.arange(hi - lo))


def test_single_process_sharded_encode_without_process_group():
    g = torch.Generator().manual_seed(1)
    tiles = torch.randn(37, 3, 4, 4, generator=g)
    out = encode_tiles_sharded(fake_encode, 37, lambda a, b: tiles[a:b], batch=8)
    assert torch.equal(out, fake_encode(tiles))


def test_single_process_passthrough():
    x = torch.randn(5, 3)
    assert all_gather_rows(x, 5) is x
    with pytest.raises(ValueError):
        all_gather_rows(x, 6)

"""Shard / all-gather logic of the multi-GPU path on the gloo backend (world_size 2 and 3, CPU)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from keep_amd.distributed import StepExchange, all_gather_rows, encode_tiles_sharded, rccl_env, shard_bounds, shard_capacity, timed_steps


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


# ------------------------------------------------------------------ bench.py's N > 1 step loop (keep_amd.distributed.StepExchange / timed_steps)
def _bench_loop_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B, D, steps = 8, 4, 7
        g = torch.Generator().manual_seed(100 + rank)                     # per-rank tiles, as bench.py draws them
        tiles = torch.randn(B, 3, 4, 4, generator=g)
        ex = StepExchange(B, D, "cpu")
        seen = []

        def step():
            f = fake_encode(tiles) + float(ex.steps)                      # the step number makes a stale buffer detectable
            seen.append(ex.submit(f))

        import time
        el = timed_steps(step, steps, ex)
        ok = ex.steps == steps and seen == [i % 2 for i in range(steps)] and el > 0
        ok = ok and all(p is None for p in ex.pending)                    # the closing fence waited for every collective
        # after the closing fence the two buffers hold the last two steps of EVERY rank, in rank order
        for back in (1, 2):
            i = (steps - back) % 2
            want = torch.cat([fake_encode(torch.randn(B, 3, 4, 4, generator=torch.Generator().manual_seed(100 + r))) + float(steps - back)
                              for r in range(world)])
            ok = ok and torch.equal(ex.gathered(i), want)
        # MAX over ranks: a slow rank sets everybody's time
        own = []
        slow = timed_steps(lambda: time.sleep(0.05 if rank == world - 1 else 0.0) or step(), 3, ex, own)
        ok = ok and slow >= 0.15 and len(own) == 1 and 0 < own[0] <= slow + 1e-6      # `own`: this rank's time before the MAX (per-rank rates)
        # every rank must run the same engine setting (each calibrates on its own at load): agreement passes, one deviating rank raises EVERYWHERE
        from keep_amd.distributed import assert_same_setting
        rows = assert_same_setting([2, 1, 8], "precision setting")
        ok = ok and len(rows) == world and all(r == [2.0, 1.0, 8.0] for r in rows)
        try:
            assert_same_setting([2, 1, 8 if rank else 6], "precision setting")
            ok = False
        except RuntimeError as e:
            ok = ok and "rank 0: [2.0, 1.0, 6.0]" in str(e)
        # ... and since a calibration's verdict can sit within a rounding of its threshold, a multi-rank job runs rank 0's plan: ranks that calibrated to
        # another plan / margin adopt it, and the agreement check then passes by construction
        from keep_amd.distributed import adopt_rank0_plan

        class FakeModel:
            def __init__(self, r):
                self.plan = [(2, 4), (4, 4), (0, 4)] if r == 0 else [(2, 4), (2, 4), (0, 4)]      # rank 0 kept one split block fewer
                self.opts = {"precision": 2.0, "label_margin": 1.1e-4 + 1e-6 * r, "strict_blocks": 0.0}
                self._label_margin_unit = 8e-5 + 1e-6 * r
            def get_plan(self): return list(self.plan)
            def set_plan(self, p): self.plan = [tuple(x) for x in p]
            def get_option(self, k): return self.opts[k]
            def set_option(self, k, v): self.opts[k] = v
            def set_precision(self, name, sb=0): self.opts["precision"] = {"fp16": 0.0, "strict": 1.0, "comp": 2.0}[name]

        fm = FakeModel(rank)
        adopted = adopt_rank0_plan(fm, device="cpu")
        ok = ok and adopted == [(2, 4), (4, 4), (0, 4)] and fm.get_plan() == adopted and abs(fm.opts["label_margin"] - 1.1e-4) < 1e-12 and abs(fm._label_margin_unit - 8e-5) < 1e-12
        rows = assert_same_setting([fm.get_option("precision")] + [float(v) for am in fm.get_plan() for v in am], "plan")
        ok = ok and len(rows) == world
        q.put((rank, bool(ok), el, slow))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_bench_step_loop_double_buffer_fence_and_max_reduce(world):
    """The loop bench.py times at N > 1: asynchronous double-buffered all-gather per step, fence (collectives + barrier) on both
    sides, elapsed = MAX over ranks -- on gloo with a stand-in encoder."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_loop_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = [q.get(timeout=5) for _ in range(world)]
    assert all(r[1] for r in res), res
    assert len({round(r[3], 9) for r in res}) == 1, "every rank must report the same (max) time"


def test_step_exchange_without_a_process_group():
    ex = StepExchange(4, 3, "cpu")
    for k in range(3):
        i = ex.submit(torch.full((4, 3), float(k)))
        assert torch.equal(ex.gathered(i), torch.full((4, 3), float(k)))
    assert timed_steps(lambda: ex.submit(torch.zeros(4, 3)), 2, ex) > 0
    from keep_amd.distributed import assert_same_setting
    assert assert_same_setting([1, 2.5]) == [[1.0, 2.5]]                 # no process group: trivially in agreement


def test_empty_slide_and_bad_arguments():
    out = encode_tiles_sharded(fake_encode, 0, lambda a, b: torch.empty(0, 3, 4, 4), batch=8, dim=4)
    assert out.shape == (0, 4)
    with pytest.raises(ValueError):
        encode_tiles_sharded(fake_encode, 0, lambda a, b: torch.empty(0, 3, 4, 4), batch=8)
    with pytest.raises(ValueError):
        encode_tiles_sharded(fake_encode, 5, lambda a, b: torch.empty(0, 3, 4, 4), batch=0)


def test_rccl_env_defaults(monkeypatch):
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    monkeypatch.delenv("MASTER_ADDR", raising=False)
    rccl_env()
    assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and os.environ["MASTER_ADDR"] == "127.0.0.1"
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "1")
    rccl_env()
    assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"          # an operator's setting wins


def test_bench_without_a_gpu_prints_no_line():
    """`python bench.py --gpus 2` on a box without GPUs: non-zero exit, a reason on stderr, no JSON line on stdout (the product has no CPU path and a
    mislabelled line is worse than none)."""
    import subprocess
    import sys
    if torch.cuda.is_available():
        pytest.skip("this is the no-GPU statement")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "MI355X" in r.stderr
    assert not any(line.lstrip().startswith("{") for line in r.stdout.splitlines())

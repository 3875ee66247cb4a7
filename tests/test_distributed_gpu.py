"""The RCCL path on the hardware that is available to the tests: ONE MI355X, backend `nccl`, world_size 1.

The sharding / un-padding logic for world_size > 1 is covered on gloo (tests/test_distributed_cpu.py); what only a GPU
can show is that the `nccl` process group initialises, that `all_gather_into_tensor` on device buffers written by the
engine's own streams is ordered correctly against them, and that bench.py's distributed leg runs end to end."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def nccl_world1():
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    from keep_amd.distributed import rccl_env
    rccl_env()                                   # dmabuf IPC (what RCCL needs across processes on these nodes) + loopback rendezvous
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def test_sharded_encode_through_rccl_matches_direct_call(nccl_world1):
    from keep_amd import KEEPModel
    from keep_amd.config import small_shape
    from keep_amd.distributed import all_gather_rows, encode_tiles_sharded
    from keep_amd.synth import synth_state_dict, synth_tiles_device
    dev = torch.device("cuda", 0)
    shape = small_shape(2, 2)
    m = KEEPModel(shape, towers=("image",))
    m.load_state_dict(synth_state_dict(shape, seed=3, text=False), strict=True)
    m.to(dev).eval()
    n = 150                                                  # ragged: 4 full batches of 32 + one of 22
    load = lambda a, b: synth_tiles_device(a, b, dev, torch.bfloat16, seed=77, unit=64)
    got = encode_tiles_sharded(m.encode_image, n, load, batch=32)
    ref = torch.cat([m.encode_image(load(a, min(a + 32, n))) for a in range(0, n, 32)])
    assert got.shape == (n, shape.projection_dim)
    assert torch.equal(got, ref)                             # same kernels, same batches: bit-identical
    assert torch.equal(all_gather_rows(ref, n), ref)
    # device-side tile generation depends on the global tile index only
    assert torch.equal(load(10, 100), torch.cat([load(10, 64), load(64, 100)]))


def test_rank0_plan_adoption_through_rccl(nccl_world1):
    """`adopt_rank0_plan` with the real engine on the `nccl` group (world size 1: the broadcast runs, the plan it hands back is the engine's own)."""
    from keep_amd import KEEPModel
    from keep_amd.config import small_shape
    from keep_amd.distributed import adopt_rank0_plan, assert_same_setting
    from keep_amd.synth import synth_state_dict
    dev = torch.device("cuda", 0)
    shape = small_shape(2, 2)
    m = KEEPModel(shape, towers=("image",))
    m.load_state_dict(synth_state_dict(shape, seed=3, text=False), strict=True)
    m.to(dev).eval()
    own, margin = m.get_plan(), m.get_option("label_margin")
    assert adopt_rank0_plan(m, device=dev) == [tuple(x) for x in own] and m.get_plan() == own and m.get_option("label_margin") == margin
    assert len(assert_same_setting([m.get_option("precision")] + [float(v) for am in m.get_plan() for v in am], "plan", device=dev)) == 1


def test_bench_distributed_leg_runs_on_nccl():
    env = dict(os.environ, KEEP_BENCH_FORCE_DIST="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))      # bench.py sets the RCCL environment itself (keep_amd.distributed.rccl_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "64",
                        "--no-cpu-baseline", "--no-breakdown", "--no-configs"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])          # the JSON line is the last thing on stdout, after RCCL's banner
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert "RCCL" in line["config"]["exchange"]
    # the N > 1 additions: every rank's own rate, and what the exchange costs on the critical path (same steps without it)
    assert len(line["per_rank_tiles_per_s"]) == 1 and line["per_rank_tiles_per_s"][0] > 0
    assert set(line["exchange"]) >= {"ms_per_step_with_exchange", "ms_per_step_encode_only", "exposed_ms_per_step", "bytes_gathered_per_step"}
    assert line["roofline"]["in_timed_region_two_lane"]["launches"] > 0


def _rccl_worker(rank, world, port, n_tiles, q):
    """One rank of the real thing: its own GPU, backend nccl (= RCCL over xGMI), the engine's encode on its shard."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from keep_amd.distributed import StepExchange, assert_same_setting, encode_tiles_sharded, rccl_env, shard_bounds, timed_steps
    rccl_env()
    import torch.distributed as dist
    from keep_amd import KEEPModel
    from keep_amd.config import small_shape
    from keep_amd.synth import synth_state_dict, synth_tiles_device
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        shape = small_shape(2, 2)
        m = KEEPModel(shape, towers=("image",))
        m.load_state_dict(synth_state_dict(shape, seed=3, text=False), strict=True)
        m.to(dev).eval()
        setting = assert_same_setting([float(v) for am in m.get_plan() for v in am], device=dev)
        load = lambda a, b: synth_tiles_device(a, b, dev, torch.bfloat16, seed=77, unit=64)
        got = encode_tiles_sharded(m.encode_image, n_tiles, load, batch=32)           # ragged shards, pipelined all-gathers
        ref = torch.cat([m.encode_image(load(a, min(a + 32, n_tiles))) for a in range(0, n_tiles, 32)])
        # a shard's batches start at the shard boundary and its last one is shorter than the single-process run's: the small-batch kernels
        # sum in another order, so rows agree to fp32 rounding (same batches -> same bits: the step loop below, and the world-1 test above)
        ok = bool((got - ref).abs().max() < 1e-5) and got.shape == ref.shape
        # bench.py's N > 1 step loop: double-buffered exchange, fences, MAX over ranks
        ex = StepExchange(32, shape.projection_dim, dev)
        lo, _ = shard_bounds(n_tiles, rank, world)
        mine = load(lo, lo + 32)
        def step():
            ex.submit(m.encode_image(mine))
        el = timed_steps(step, 4, ex)
        last = ex.gathered((ex.steps - 1) % 2)
        for r in range(world):
            rlo, _ = shard_bounds(n_tiles, r, world)
            ok = ok and bool(torch.equal(last[r * 32:(r + 1) * 32], m.encode_image(load(rlo, rlo + 32))))
        q.put((rank, ok, el > 0, len(setting)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_tiles", [150, 97])
def test_two_ranks_over_rccl_match_the_single_process_result(n_tiles):
    """Needs two MI355X in one node (skipped on the one-GPU test box): two processes, one per GPU, backend nccl.  The sharded encode with
    ragged shards and asynchronous all-gathers returns, on every rank, exactly the rows a single process computes; the bench's step loop
    (StepExchange / timed_steps) delivers every rank's batch to every rank."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs in the node (RCCL between processes)")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_worker, args=(r, 2, port, n_tiles, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(2))
    assert [r[0] for r in res] == [0, 1] and all(r[1] and r[2] and r[3] == 2 for r in res), res


def test_bench_refuses_more_gpus_than_the_node_has():
    """`python bench.py --gpus N` as the driver invokes it (no torchrun around it): with fewer than N GPUs visible it must exit non-zero with a clear
    message and print NO JSON line -- never a 1-GPU figure labelled as N."""
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0
    assert f"--gpus {n}" in r.stderr and "visible" in r.stderr
    assert not any(line.lstrip().startswith("{") for line in r.stdout.splitlines())


def test_bench_starts_its_own_ranks():
    """The self-spawn path on the one GPU the test box has (KEEP_BENCH_FORCE_SPAWN=1 makes `--gpus 1` take it): bench.py re-executes under
    torch.distributed.run, the rank initialises RCCL, runs the step loop with the all-gather and the configs-4/5 multi-GPU slide leg, and the parent's
    stdout ends with ONE JSON line whose n_gpus is the number of ranks RCCL saw."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["KEEP_BENCH_FORCE_SPAWN"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-breakdown",
                        "--no-configs", "--no-sustained", "--slide", "--slide-tiles", "1500"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["rccl_ranks_seen"] == 1 and line["value"] > 0 and "RCCL" in line["config"]["exchange"]
    sl = line["slide"]
    assert sl["tiles"] == 1500 and sl["ranks"] == 1 and sl["tiles_per_rank"] == 1500 and sl["prob_map_rows_this_rank"] == 1500
    assert sl["every_rank_same_embeddings_label_ratio"] and 0 <= sl["slide_label"] < 4 and 0.0 <= sl["tumour_ratio"] <= 1.0

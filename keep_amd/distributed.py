"""Multi-GPU sharding of a slide's tiles: one process per GPU, contiguous tile shards,
no data-path collective inside the encoders, and an all-gather of the per-tile embeddings for the
slide-level steps that need every tile (prompt screening means, spatial refine).

Works with any initialised ``torch.distributed`` backend: ``nccl`` (= RCCL over xGMI on MI355X) for
GPU tensors, ``gloo`` for the CPU tests.  The reference has no inference-side multi-GPU code
(WSI scripts pin ``device='cuda:0'``, e.g. zeroshot_subtyping_WSI.py:26); this is new.

The exchange is pipelined: every rank encodes its shard in batches of ``batch`` tiles, and the all-gather of
batch j (``[batch, D]`` per rank, 0.79 MB at 256 x 768 fp32) is issued asynchronously while batch j+1 is being
encoded, so the collective never sits on the critical path; the gathered ``[n_batches, world, batch, D]`` slab is
un-padded into ``[n_tiles, D]`` with ONE index_select (no per-rank Python copies).
"""
from __future__ import annotations

import os
import time
from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def rccl_env() -> None:
    """Environment a multi-process RCCL job needs on this platform; call before the first HIP call of the process.

    ``HSA_ENABLE_IPC_MODE_LEGACY=0``: the host driver of the MI355X nodes this was built on only supports dmabuf-based IPC handles.
    RCCL's intra-node transport (and torch's CUDA-tensor sharing) exchange device buffers between the ranks' processes through
    ``hipIpcGetMemHandle``; in the legacy IPC mode that call fails with ``invalid argument`` as soon as there is a second process
    (world size 1 never opens a peer handle, which is why single-GPU runs work without it).  ``setdefault``: an operator's own
    setting wins.  The rendezvous address defaults to loopback: container host names do not always resolve."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition; the first ``n_items % world`` ranks get one extra item."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"rank {rank} / world {world}")
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_capacity(n_items: int, world: int) -> int:
    return -(-n_items // world)


def _owner_and_offset(n_total: int, world: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """For every global row: the rank that holds it under ``shard_bounds`` and its offset inside that shard."""
    g = torch.arange(n_total, device=device, dtype=torch.int64)
    q, rem = divmod(n_total, world)
    head = rem * (q + 1)                                  # rows held by the ranks that carry one extra item
    rank = torch.where(g < head, g // (q + 1), rem + (g - head) // max(q, 1))
    lo = rank * q + torch.clamp(rank, max=rem)
    return rank, g - lo


def all_gather_rows(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """Gather row-sharded ``local`` ([n_local, D], partition = shard_bounds) into [n_total, D] on every rank.

    Shards are padded to the common capacity so a single ``all_gather_into_tensor`` moves everything; the
    padding is dropped with one index_select.
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
    if hi - lo == cap:
        send = local.contiguous()
    else:
        send = local.new_zeros((cap, D))
        send[: hi - lo] = local
    recv = local.new_empty((world * cap, D))
    dist.all_gather_into_tensor(recv, send, group=group)
    if n_total == world * cap:
        return recv
    owner, off = _owner_and_offset(n_total, world, local.device)
    return recv.index_select(0, owner * cap + off)


def encode_tiles_sharded(encode: Callable[[torch.Tensor], torch.Tensor], n_tiles: int,
                         load_tiles: Callable[[int, int], torch.Tensor], batch: int = 256, group=None,
                         dim: Optional[int] = None, device=None) -> torch.Tensor:
    """Each rank encodes tiles [lo, hi) in batches with ``encode`` (e.g. KEEPModel.encode_image) and all
    ranks receive the full [n_tiles, D] embedding matrix.  An empty slide (``n_tiles == 0``) has no tile to learn D from:
    pass ``dim`` (and ``device``) to get the empty ``[0, dim]`` matrix, otherwise it is an error."""
    if n_tiles < 0 or batch < 1:
        raise ValueError(f"n_tiles {n_tiles}, batch {batch}")
    if n_tiles == 0:
        if dim is None:
            raise ValueError("n_tiles == 0: pass dim= (the embedding width cannot be learnt from an empty slide)")
        return torch.empty((0, dim), dtype=torch.float32, device=device)
    distributed = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if distributed else 0
    world = dist.get_world_size(group) if distributed else 1
    lo, hi = shard_bounds(n_tiles, rank, world)
    cap = shard_capacity(n_tiles, world)
    nb = max(1, -(-cap // batch))                         # every rank issues the same number of collectives
    slab = None                                           # [nb, world, batch, D], allocated once D / dtype / device are known
    pending = []
    for j in range(nb):
        a, b = min(lo + j * batch, hi), min(lo + (j + 1) * batch, hi)
        if b > a:
            f = encode(load_tiles(a, b))
        elif slab is None:
            f = encode(load_tiles(0, 1))[:0]              # empty shard: only to learn D / dtype / device
        else:
            f = slab.new_empty((0, slab.shape[-1]))
        if slab is None:
            slab = f.new_empty((nb, world, batch, f.shape[1]))
        if not distributed:
            slab[j, 0, : b - a] = f
            continue
        if b - a == batch:
            send = f.contiguous()
        else:
            send = f.new_zeros((batch, f.shape[1]))
            send[: b - a] = f
        # async: the collective of batch j runs while batch j+1 is encoded; `send` stays referenced until waited for
        pending.append((dist.all_gather_into_tensor(slab[j].view(world * batch, -1), send, group=group, async_op=True), send))
    for work, _ in pending:
        work.wait()
    owner, off = _owner_and_offset(n_tiles, world, slab.device)
    pos = (off // batch) * (world * batch) + owner * batch + off % batch
    return slab.view(nb * world * batch, -1).index_select(0, pos)


# ------------------------------------------------------------------------------------------------
# The benchmark's step loop (bench.py, N > 1): encode one batch per step, exchange it, time exactly n steps between fences.
# ------------------------------------------------------------------------------------------------
class StepExchange:
    """Double-buffered asynchronous all-gather of one ``[batch, dim]`` result per step: the collective of step i runs while step
    i+1 is being encoded; buffer i % depth is reused only after its previous collective has been waited for.  Backend-agnostic
    (``nccl`` = RCCL on the GPUs, ``gloo`` in the CPU tests); without a process group it degenerates to keeping the last results."""

    def __init__(self, batch: int, dim: int, device, dtype=torch.float32, group=None, depth: int = 2):
        self.group = group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.device = torch.device(device)
        self.buffers: List[torch.Tensor] = [torch.empty(self.world * batch, dim, device=self.device, dtype=dtype) for _ in range(depth)]
        self.pending: List[Optional[tuple]] = [None] * depth
        self.steps = 0

    def submit(self, f: torch.Tensor) -> int:
        """Queue the exchange of this step's result; returns the index of the buffer it lands in."""
        i = self.steps % len(self.buffers)
        self.steps += 1
        if self.pending[i] is not None:
            self.pending[i][0].wait()              # stream-level wait on GPUs: frees buffer i (and keeps the old `f` alive until then)
        if self.distributed:
            self.pending[i] = (dist.all_gather_into_tensor(self.buffers[i], f, group=self.group, async_op=True), f)
        else:
            self.buffers[i].copy_(f)
        return i

    def gathered(self, i: int) -> torch.Tensor:
        if self.pending[i] is not None:
            self.pending[i][0].wait()
        return self.buffers[i]

    def fence(self) -> None:
        """Every queued collective done on every rank, every device idle: both ends of a timed region."""
        for k, h in enumerate(self.pending):
            if h is not None:
                h[0].wait()
                self.pending[k] = None
        if self.distributed:
            dist.barrier(group=self.group)
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)


def timed_steps(step: Callable[[], None], n_steps: int, exchange: StepExchange, own: Optional[list] = None) -> float:
    """Seconds for exactly ``n_steps`` calls of ``step`` between two fences (barrier + device synchronisation), MAX over the ranks.
    ``own``: a list that receives this rank's own time (before the MAX), for per-rank rates."""
    exchange.fence()
    t0 = time.perf_counter()
    for _ in range(n_steps):
        step()
    exchange.fence()
    el = time.perf_counter() - t0
    if own is not None:
        own.append(el)
    if exchange.distributed:
        t = torch.tensor([el], device=exchange.device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=exchange.group)
        el = float(t.item())
    return el


def assert_same_setting(values, what: str = "precision setting", device=None, group=None) -> list:
    """Every rank must run the SAME engine setting for a multi-rank figure to mean anything (each rank calibrates on its own at load):
    all-gather ``values`` (a short sequence of numbers) and raise on every rank if any rank differs.  Returns the gathered rows."""
    row = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    if not (dist.is_available() and dist.is_initialized()):
        return [row.tolist()]
    world = dist.get_world_size(group)
    rows = [torch.empty_like(row) for _ in range(world)]
    dist.all_gather(rows, row, group=group)
    got = [r.tolist() for r in rows]
    if any(r != got[0] for r in got):
        raise RuntimeError(f"ranks disagree on the {what}: " + "; ".join(f"rank {i}: {r}" for i, r in enumerate(got)))
    return got


def adopt_rank0_plan(model, device=None, group=None, src: int = 0) -> list:
    """Every rank calibrates on its own at load, and the rule's verdict on a candidate plan can sit within a rounding of its threshold (two boxes have been
    seen to keep plans that differ in ONE block): a multi-rank job then runs rank ``src``'s plan everywhere.  The plan is a property of the weights -- identical
    on every rank -- and was verified on that rank against the same probe; what is broadcast is (precision, label margin figures, the per-block plan).  Returns
    the adopted plan.  Without a process group: a no-op."""
    plan = [tuple(int(v) for v in am) for am in model.get_plan()]
    if not (dist.is_available() and dist.is_initialized()):
        return plan
    unit = getattr(model, "_label_margin_unit", None)
    row = torch.tensor([float(model.get_option("precision")), float(model.get_option("label_margin")), float(unit or 0.0)] + [float(v) for am in plan for v in am],
                       dtype=torch.float64, device=device)
    dist.broadcast(row, src=src, group=group)
    got = row.tolist()
    new_plan = [(int(round(got[3 + 2 * i])), int(round(got[4 + 2 * i]))) for i in range(len(plan))]
    if dist.get_rank(group) != src:
        precision = {0: "fp16", 1: "strict", 2: "comp"}[int(round(got[0]))]
        if int(round(model.get_option("precision"))) != int(round(got[0])):
            model.set_precision(precision, int(model.get_option("strict_blocks")))
        if new_plan != plan:
            model.set_plan(new_plan)
        model.set_option("label_margin", got[1])
        model._label_margin_unit = got[2] or None
    return new_plan

#!/usr/bin/env python3
"""Headline benchmark: 224x224 tiles encoded per second through the ViT-L/16 image tower
(BASELINE.json configs[1]: batch 256 synthetic bf16 tiles per GPU, random-init weights).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = KEEPModel.encode_image on one batch of 256 device-resident tiles per rank (+ the RCCL
all-gather of the [256,768] embeddings when N > 1: the slide-level pooling exchange of config 4).
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import platform
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from keep_amd import KEEPModel, PROFILE_TAGS, vit_flops_per_tile          # noqa: E402
from keep_amd.config import KEEPShape                                      # noqa: E402
from keep_amd.synth import synth_state_dict, synth_tiles                   # noqa: E402

PEAK_F16_TFLOPS = 2516.6     # 256 CU x 4096 FLOP/clk x 2.4 GHz, dense
DOMINANT_TAG = "vit.fc1"     # gemm_f16_nt_kernel<EPI_GELU_F16>: [B*197,1024] x [1024,4096], 32 % of the FLOPs


T_START = time.perf_counter()


def log(msg: str) -> None:
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.perf_counter() - T_START:6.1f}s] {msg}", file=sys.stderr, flush=True)


def usable_cpus() -> int:
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_model_name() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def cpu_baseline(sd, tiles: int = 8, iters: int = 40, min_seconds: float = 10.0):
    """The oracle (fp32 torch-CPU restatement of the reference's encode_image) on the host cores."""
    from oracle import keep_oracle as O
    torch.set_num_threads(min(usable_cpus(), 64))
    x = synth_tiles(tiles, seed=0)
    with torch.no_grad():
        O.encode_image(sd, x[:1])                                   # warm-up
        t0 = time.perf_counter()
        done = 0
        for _ in range(iters):
            O.encode_image(sd, x)
            done += 1
            if time.perf_counter() - t0 > min_seconds:              # bounded sample: ~10 s of CPU work
                break
        dt = time.perf_counter() - t0
    iters = done
    return {"value": round(tiles * iters / dt, 3), "unit": "tiles/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{iters} x {tiles} synthetic 224x224 tiles, fp32 torch-CPU restatement (oracle/keep_oracle.py) "
                      f"of KEEPModel.encode_image, same synthetic weights; CPU: {cpu_model_name()}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="tiles per GPU per step")
    ap.add_argument("--precision", default="comp", choices=["comp", "fp16", "strict"])
    ap.add_argument("--pixel-dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--opt", action="append", default=[], help="engine option name=value (e.g. gemm_impl=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: keep_amd has no CPU execution path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    use_dist = world > 1 or os.environ.get("KEEP_BENCH_FORCE_DIST") == "1"     # the override exercises the RCCL path on one GPU
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    log(f"start: world={world} cpus={usable_cpus()} (os.cpu_count={os.cpu_count()})")
    torch.set_num_threads(min(usable_cpus(), 16))
    shape = KEEPShape()
    sd = synth_state_dict(shape, seed=0, text=False)                 # identical weights on every rank
    model = KEEPModel(shape, precision=args.precision)
    model.load_state_dict(sd, strict=True)
    model.to(dev).eval()
    for kv in args.opt:
        k, v = kv.split("=")
        model.set_option(k, float(v))
    model.reserve(tiles=args.batch)
    log("weights uploaded, workspace reserved")

    B = args.batch
    pix = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.pixel_dtype]
    g = torch.Generator(device=dev).manual_seed(1234 + rank)        # per-rank tiles, generated on device
    tiles = torch.randn(B, 3, 224, 224, device=dev, generator=g, dtype=torch.float32).to(pix)
    # double-buffered so the RCCL all-gather of step i overlaps the encode of step i+1
    gathered = [torch.empty(world * B, shape.projection_dim, device=dev, dtype=torch.float32) for _ in range(2)] if use_dist else None
    pending = [None, None]
    step_no = [0]

    def step():
        f = model.encode_image(tiles)
        if use_dist:
            i = step_no[0] & 1
            if pending[i] is not None:
                pending[i][0].wait()                       # stream-level wait, frees buffer i (and keeps f alive until then)
            pending[i] = (dist.all_gather_into_tensor(gathered[i], f, async_op=True), f)
            step_no[0] += 1
        return f

    def fence():
        if use_dist:
            for h in pending:
                if h is not None:
                    h[0].wait()
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    log("warm-up done")
    model.profile_enable(DOMINANT_TAG)
    model.profile_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    log(f"timed region done: {elapsed / args.steps * 1e3:.2f} ms/step")
    dom_ms, dom_n, dom_flops = model.profile_read(DOMINANT_TAG)
    model.profile_disable()
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # second, untimed pass on ONE internal stream: clean per-kernel times (lanes do not overlap) for the
    # breakdown and for the dominant kernel in isolation
    breakdown, iso = None, None
    if not args.no_breakdown and rank == 0:
        streams_was = model._options.get("streams", 2)
        model.set_option("streams", 1)
        model.profile_enable(None)
        model.profile_reset()
        for _ in range(2):
            model.encode_image(tiles)
        torch.cuda.synchronize(dev)
        breakdown = {}
        for tag in PROFILE_TAGS:
            ms, n, fl = model.profile_read(tag)
            if n:
                breakdown[tag] = round(ms / 2, 3)
            if tag == DOMINANT_TAG and n:
                iso = {"achieved": round(fl / (ms * 1e-3) / 1e12, 1), "avg_launch_ms": round(ms / n, 4), "launches": n}
        model.profile_disable()
        model.set_option("streams", streams_was)

    # untimed: the same 256-tile call with 8 tiles whose fp32-oracle features are a committed fixture
    # (tests/golden/vit_d24_bench.npz: same seed-0 weights; generated by tools/make_golden.py, pinned against Dinov2Model)
    parity = None
    gpath = os.path.join(ROOT, "tests", "golden", "vit_d24_bench.npz")
    if rank == 0 and os.path.exists(gpath) and B >= 8:
        import numpy as np
        g = np.load(gpath)
        gt = synth_tiles(int(g["batch"]), seed=int(g["tile_seed"])).to(dev)
        batch = torch.randn(B, 3, 224, 224, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
        batch[: gt.shape[0]] = gt
        got = model.encode_image(batch)[: gt.shape[0]].cpu()
        ref = torch.from_numpy(g["features"])
        bank = torch.nn.functional.normalize(torch.randn(64, ref.shape[1], generator=torch.Generator().manual_seed(3)), dim=-1)
        d = (got @ bank.t() - ref @ bank.t()).abs()
        parity = {"max_abs_dcos": float(f"{d.max().item():.3e}"), "rms_dcos": float(f"{d.pow(2).mean().sqrt().item():.3e}"),
                  "n_cosines": int(d.numel()),
                  # BASELINE.json's second metric: zero-shot sim match-rate vs ref
                  "sim_match_rate_1e-4": round(float((d <= 1e-4).float().mean().item()), 4),
                  "argmax_match_rate": round(float(((got @ bank.t()).argmax(1) == (ref @ bank.t()).argmax(1)).float().mean().item()), 4),
                  "north_star_tolerance": 1e-4, "reference": "tests/golden/vit_d24_bench.npz (fp32 oracle features of 8 tiles, same weights)",
                  "note": "fp16 MFMA operands sit on the 1e-4 budget (DESIGN.md section 5); --precision strict is 200x inside it"}

    if rank == 0:
        tiles_per_s = world * B * args.steps / elapsed
        avg_ms = dom_ms / max(dom_n, 1)
        achieved = dom_flops / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0     # executed FLOPs / summed launch time
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")         # rocprofv3 --pmc passes (see profiles/README.md)
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(DOMINANT_TAG, {}).get("bytes_per_launch")
            except (OSError, ValueError):
                traffic = None
        line = {
            "metric": "224x224 tiles encoded/sec (whole node)", "value": round(tiles_per_s, 2), "unit": "tiles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"fp16": "fp16", "comp": "fp16+mxfp4", "strict": "fp16x3"}[args.precision],
            "data": "synthetic (randn tiles generated on device, seeded random-init weights)",
            "config": {"workload": "ViT-L/16 image encoder only (KEEP encode_image), batch 256 synthetic 224x224 "
                                   f"{args.pixel_dtype} tiles per GPU, 1xMI355X per rank",
                       "tiles_per_gpu_per_step": B, "precision": args.precision,
                       "exchange": "RCCL all_gather of [256,768] fp32 embeddings per step" if world > 1 else "none",
                       "mfma_frac_end_to_end": round(tiles_per_s / world * vit_flops_per_tile() / (PEAK_F16_TFLOPS * 1e12), 4)},
            "roofline": {"bound": "mfma", "kernel": "keepk::gemm_f16_v2_kernel<256,2,4,4,EPI_GELU_F16> (vit.fc1)",
                         "achieved": round(achieved, 1), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_F16_TFLOPS, 4), "traffic": traffic,
                         "avg_launch_ms": round(avg_ms, 4), "launches": dom_n,
                         "flops_per_launch": round(dom_flops / max(dom_n, 1), 1),
                         "note": "timed region runs 2 sub-batches on 2 internal streams: a launch shares the GPU with the other "
                                 "lane's kernels, so its duration (and this fraction) is lower than in isolation"},
        }
        if iso is not None:
            iso["frac"] = round(iso["achieved"] / PEAK_F16_TFLOPS, 4)
            line["roofline_isolated"] = iso
        if parity is not None:
            line["parity"] = parity
        if breakdown is not None:
            line["breakdown_ms_per_step_single_stream"] = breakdown
        if world == 1 and not args.no_cpu_baseline:
            log("cpu baseline (oracle on host cores) ...")
            line["cpu_baseline"] = cpu_baseline(sd)
            log("cpu baseline done")
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Interleaved A/B of engine OPTIONS on one box and one library: 256-tile encode steps under each option set, several rounds, plus the per-tag
HIP-event times of one single-stream pass per arm (so a change is visible in the operator it touches).  The part runs at its power cap: only
same-box, same-minute comparisons mean anything.

    python tools/ab_options.py --arm base --arm proj_impl=2128 --arm cls_qkv=0 [--plan 1,8 | --calibrate] [--steps 20] [--rounds 3]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from keep_amd import KEEPModel, PROFILE_TAGS                              # noqa: E402
from keep_amd.config import KEEPShape                                     # noqa: E402
from keep_amd.model import plan_string, prefix_plan                       # noqa: E402
from keep_amd.synth import synth_state_dict                               # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arm", action="append", default=[], help="'base' or comma-separated name=value engine options")
    ap.add_argument("--plan", default="1,8", help="prefix plan full,mlp (ignored with --calibrate)")
    ap.add_argument("--calibrate", action="store_true")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    shape = KEEPShape()
    m = KEEPModel(shape)
    m.auto_calibrate = args.calibrate
    m.load_state_dict(synth_state_dict(shape, seed=0))
    m.to(dev).eval()
    if not args.calibrate:
        full, mlp = (int(v) for v in args.plan.split(","))
        m.set_plan(prefix_plan(shape.vision.depth, full, mlp))
    print("plan:", plan_string(m.get_plan()), flush=True)
    m.reserve(tiles=256)
    g = torch.Generator(device=dev).manual_seed(1234)
    tiles = torch.randn(256, 3, 224, 224, device=dev, generator=g, dtype=torch.float32).to(torch.bfloat16)
    arms = args.arm or ["base"]
    defaults = {}

    def apply(arm):
        for k, v in defaults.items():
            m.set_option(k, v)
        if arm != "base":
            for kv in arm.split(","):
                k, v = kv.split("=")
                defaults.setdefault(k, m.get_option(k))
                m.set_option(k, float(v))

    for arm in arms:                          # remember every touched option's default before the first timing
        apply(arm)
    feats, rates = {}, {a: [] for a in arms}
    for r in range(args.rounds):
        for arm in (arms[r % len(arms):] + arms[:r % len(arms)]):          # rotate the order from round to round: a thermal drift within a round cancels

            apply(arm)
            for _ in range(3):
                f = m.encode_image(tiles)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                f = m.encode_image(tiles)
            torch.cuda.synchronize(dev)
            rates[arm].append(256 * args.steps / (time.perf_counter() - t0))
            feats[arm] = f.clone()
    for arm in arms:
        apply(arm)
        m.set_option("streams", 1)
        m.profile_enable(None)
        m.profile_reset()
        for _ in range(3):
            m.encode_image(tiles)
        torch.cuda.synchronize(dev)
        tags = {t: round(m.profile_read(t)[0] / 3, 3) for t in PROFILE_TAGS if m.profile_read(t)[1]}
        m.profile_disable()
        m.set_option("streams", 2)
        d = (feats[arm] - feats[arms[0]]).abs().max().item()
        print(f"{arm:40s} tiles/s {' '.join(f'{x:7.1f}' for x in rates[arm])}  (mean {sum(rates[arm]) / len(rates[arm]):7.1f})  max|dfeat| vs first arm {d:.2e}")
        print(f"{'':40s} single-stream ms/step per tag: {tags}", flush=True)


if __name__ == "__main__":
    main()

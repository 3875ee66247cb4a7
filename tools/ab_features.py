#!/usr/bin/env python3
"""A/B helper: encode a fixed batch with the library named by KEEP_HIP_LIB and save the features (bit-comparison between builds that must not
change results), plus a quick timing.   KEEP_HIP_LIB=keep_amd/libX.so python tools/ab_features.py --out gpurun_out/x.pt
python tools/ab_features.py --compare a.pt b.pt"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out")
    ap.add_argument("--compare", nargs=2)
    ap.add_argument("--tiles", type=int, default=256)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--precision", default="comp")
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--breakdown", action="store_true", help="also print the per-operator times of a single-stream pass")
    args = ap.parse_args()
    if args.compare:
        a, b = (torch.load(p) for p in args.compare)
        print(f"bit-equal: {bool(torch.equal(a, b))}; max abs diff {float((a - b).abs().max()):.3e}")
        return
    from keep_amd import KEEPModel, _lib
    from keep_amd.config import KEEPShape
    from keep_amd.synth import synth_state_dict
    dev = torch.device("cuda", 0)
    m = KEEPModel(KEEPShape(), precision=args.precision, towers=("image",))
    m.auto_calibrate = False
    m.load_state_dict(synth_state_dict(KEEPShape(), seed=0, text=False))
    m.to(dev).eval()
    for kv in args.opt:
        k, v = kv.split("=")
        m.set_option(k, float(v))
    g = torch.Generator(device=dev).manual_seed(1234)
    x = torch.randn(args.tiles, 3, 224, 224, device=dev, generator=g).to(torch.bfloat16)
    for _ in range(3):
        f = m.encode_image(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        f = m.encode_image(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(f"{os.path.basename(_lib.LIB_PATH)}: {args.tiles / dt:.1f} tiles/s, {dt * 1e3:.3f} ms/step")
    if args.breakdown:
        from keep_amd import PROFILE_TAGS
        m.set_option("streams", 1)
        m.profile_enable(None)
        m.profile_reset()
        for _ in range(3):
            m.encode_image(x)
        torch.cuda.synchronize()
        print("   single stream, ms per step:", {t: round(m.profile_read(t)[0] / 3, 3) for t in PROFILE_TAGS if m.profile_read(t)[1]})
        m.profile_disable()
    if args.out:
        torch.save(f.cpu(), args.out)


if __name__ == "__main__":
    main()

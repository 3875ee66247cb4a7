#!/usr/bin/env python3
"""How much head-room does the default ('comp') mode keep under the 1e-4 cosine tolerance on tiles the committed oracle fixture
does not cover?  The strict mode (4e-7 from the fp32 oracle, tests/test_towers_gpu.py) serves as the reference here: N chunks of
256 fresh tiles x 64 prompts each, max / rms |dcos| of the mode under test, per chunk and overall.

    python tools/precision_margin.py [--chunks 16] [--opt comp_mlp_blocks=8 ...]
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keep_amd import KEEPModel
from keep_amd.config import KEEPShape
from keep_amd.synth import synth_prompts, synth_state_dict, synth_tiles_device

ap = argparse.ArgumentParser()
ap.add_argument("--chunks", type=int, default=16)
ap.add_argument("--seed", type=int, default=900_000)
ap.add_argument("--weight-seed", type=int, default=0)
ap.add_argument("--opt", action="append", default=[])
args = ap.parse_args()
dev = torch.device("cuda:0")
sd = synth_state_dict(KEEPShape(), seed=args.weight_seed)
ref_m = KEEPModel(precision="strict"); ref_m.load_state_dict(sd); ref_m.to(dev)
m = KEEPModel(); m.load_state_dict(sd); m.to(dev)
for kv in args.opt:
    k, v = kv.split("="); m.set_option(k, float(v))
toks = {k: v.to(dev) for k, v in synth_prompts(64, 256, seed=args.seed).items()}
ref_t, got_t = ref_m.encode_text(toks), m.encode_text(toks)
worst, sq, n = 0.0, 0.0, 0
for c in range(args.chunks):
    x = synth_tiles_device(0, 256, dev, torch.float32, seed=args.seed + c)
    d = (m.encode_image(x) @ got_t.t() - ref_m.encode_image(x) @ ref_t.t()).abs()
    worst = max(worst, d.max().item()); sq += d.pow(2).sum().item(); n += d.numel()
    print(f"chunk {c:3d}: max|dcos| {d.max().item():.3e}  rms {d.pow(2).mean().sqrt().item():.3e}", flush=True)
print(f"{' '.join(args.opt) or 'defaults'}: {n} cosines, max|dcos| {worst:.3e}, rms {(sq / n) ** 0.5:.3e}, head-room to 1e-4: {(1 - worst / 1e-4) * 100:.0f} %")

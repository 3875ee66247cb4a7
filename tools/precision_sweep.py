#!/usr/bin/env python3
"""Accuracy / throughput of the precision modes on the full-depth image tower (synthetic weights):
max |dcos| of 16 tiles x 64 random unit text vectors vs the fp32 CPU oracle, and tiles/s at batch 256."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keep_amd import KEEPModel
from keep_amd.config import KEEPShape
from keep_amd.synth import synth_state_dict, synth_tiles
from oracle import keep_oracle as O

torch.set_num_threads(16)
sd = synth_state_dict(KEEPShape(), seed=0, text=False)
x = synth_tiles(16, seed=77)
g = torch.Generator().manual_seed(3)
txt = torch.nn.functional.normalize(torch.randn(64, 768, generator=g), dim=-1)
with torch.no_grad():
    ref = O.encode_image(sd, x)
from keep_amd.synth import towers_of
m = KEEPModel(towers=towers_of(sd))
m.load_state_dict(sd)
m.to("cuda:0")
big = torch.randn(256, 3, 224, 224, device="cuda").to(torch.bfloat16)
for name, prec, sb in (("fp16", "fp16", 0), ("strict_blocks=2", "fp16", 2), ("strict_blocks=4", "fp16", 4), ("strict_blocks=6", "fp16", 6),
                       ("strict_blocks=12", "fp16", 12), ("strict", "strict", 0)):
    m.set_precision(prec, sb)
    out = m.encode_image(x)
    d = (out @ txt.t() - ref @ txt.t()).abs()
    for _ in range(2):
        m.encode_image(big)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        m.encode_image(big)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{name:18s} max|dcos| {d.max():.2e}  rms {d.pow(2).mean().sqrt():.2e}  ||df||max {(out - ref).norm(dim=-1).max():.2e}   {256 / dt:7.0f} tiles/s")

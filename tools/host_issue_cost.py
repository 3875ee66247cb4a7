#!/usr/bin/env python3
"""Host time to ISSUE one 256-tile encode_image call (launches only, no synchronisation) against its GPU time: how much head-room a
process-per-GPU deployment has before the host thread becomes the limit (measured: 1.6 ms of a 38.7 ms step)."""
import time, torch, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keep_amd.config import KEEPShape
from keep_amd.synth import synth_state_dict
from keep_amd.model import KEEPModel
shape = KEEPShape()
m = KEEPModel(shape, precision="comp", towers=("image",)); m.load_state_dict(synth_state_dict(shape, seed=0, text=False), strict=True); m.to("cuda").eval()
x = torch.randn(256, 3, 224, 224, device="cuda").to(torch.bfloat16)
for _ in range(3): m.encode_image(x)
torch.cuda.synchronize()
t = []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter(); m.encode_image(x); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    t.append((t1 - t0, t2 - t0))
print("host issue ms", [round(a * 1e3, 2) for a, _ in t]); print("total ms", [round(b * 1e3, 2) for _, b in t])

#!/usr/bin/env python3
"""Effective shader clock with and without the encoder running (keep_clock_probe: s_memtime against the 100 MHz s_memrealtime;
tools/ubench/clock_probe.hip checked both counters against HIP-event time: 2 395 MHz and 100.0 MHz on an idle MI355X)."""
import sys, torch
sys.path.insert(0, '.')
from keep_amd import KEEPModel
from keep_amd.config import KEEPShape
from keep_amd.synth import synth_state_dict
mm = KEEPModel(towers=("image",)); mm.auto_calibrate = False
mm.load_state_dict(synth_state_dict(KEEPShape(), seed=0, text=False)); mm.to("cuda:0")
buf = torch.zeros(64, 2, dtype=torch.int64, device="cuda"); torch.cuda.synchronize()
mhz = lambda rows: [round(100.0 * a / b) for a, b in rows.tolist() if b]
for i in range(3):
    mm.clock_probe(buf[i], 1000)
torch.cuda.synchronize(); print("idle:", mhz(buf[:3]))
x = torch.randn(256, 3, 224, 224, device="cuda").to(torch.bfloat16)
for _ in range(30): mm.encode_image(x)
side = torch.cuda.Stream()
for i in range(3, 43):
    mm.encode_image(x); mm.clock_probe(buf[i], 300, side)
torch.cuda.synchronize(); print("under the encoder (side stream, one probe per 256-tile step):", mhz(buf[3:43]))

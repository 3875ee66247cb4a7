#!/usr/bin/env python3
"""Does the patch-embedding GEMM need its split product?  Probe rms (256 tiles x 64 prompts and isotropic) of the calibrated plan with the patch embedding as a
split product (default) and as one fp16 pass, and the step time of both (interleaved)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keep_amd import KEEPModel
from keep_amd.config import KEEPShape
from keep_amd.model import plan_string
from keep_amd.synth import synth_prompts, synth_state_dict

dev = torch.device("cuda", 0)
m = KEEPModel(KEEPShape())
m.load_state_dict(synth_state_dict(KEEPShape(), seed=0))
m.to(dev).eval()
plan = m.get_plan()
print("plan:", plan_string(plan), "probe rms", m.calibration["probe_rms_dcos"])
g = torch.Generator(device=dev).manual_seed(4242)
tiles = torch.randn(256, 3, 224, 224, device=dev, generator=g, dtype=torch.float32).to(torch.bfloat16)
bank = m.encode_text({k: v.to(dev) for k, v in synth_prompts(64, 256, seed=1).items()})
m.set_precision("strict"); ref_f = m.encode_image(tiles); ref = m.similarity(ref_f, bank); m.set_precision("comp"); m.set_plan(plan)
for ps in (1, 0, 1, 0):
    m.set_option("patch_split", ps)
    f = m.encode_image(tiles)
    d = m.similarity(f, bank) - ref
    for _ in range(3): m.encode_image(tiles)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): m.encode_image(tiles)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
    print(f"patch_split={ps}: rms vs bank {float(d.pow(2).mean().sqrt()):.3e} max {float(d.abs().max()):.3e} isotropic {float((f - ref_f).pow(2).sum(1).mean().div(768).sqrt()):.3e}  {ms:.3f} ms/step {256e3 / ms:.0f} tiles/s")

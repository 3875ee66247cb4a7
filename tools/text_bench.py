#!/usr/bin/env python3
"""encode_text throughput (BASELINE config 3 shape: 64 prompts x 256 tokens) and the dual-tower config-3 run."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keep_amd import KEEPModel, PROFILE_TAGS, bert_flops_per_prompt
from keep_amd.config import KEEPShape
from keep_amd.synth import synth_prompts, synth_state_dict

sd = synth_state_dict(KEEPShape(), seed=0)
from keep_amd.synth import towers_of
m = KEEPModel(towers=towers_of(sd)); m.load_state_dict(sd); m.to("cuda:0")
def flops_at(T):       # 12 layers at sequence length T + pooler (SURVEY.md §8d formula, padded length = 256)
    H, I, L = 768, 3072, 12
    return L * (2 * T * H * 3 * H + 4 * T * T * H + 2 * T * H * H + 4 * T * H * I) + 2 * H * H
assert flops_at(256) == bert_flops_per_prompt()
for trim in (False, True):
    m.trim_padding = trim
    for P in (1, 8, 64, 256):
        toks = {k: v.cuda() for k, v in synth_prompts(P, 256, seed=1).items()}
        for _ in range(3): m.encode_text(toks)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 20 if P <= 64 else 5
        for _ in range(n): m.encode_text(toks)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        T = m.last_text_length
        print(f"encode_text P={P:4d} padded T=256, run at T={T:3d}: {dt*1e3:8.3f} ms  {P/dt:9.1f} prompts/s  "
              f"{P*bert_flops_per_prompt()/dt/1e12:7.1f} TFLOP/s padded-equivalent, {P*flops_at(T)/dt/1e12:7.1f} TFLOP/s executed")
m.trim_padding = False
toks = {k: v.cuda() for k, v in synth_prompts(64, 256, seed=1).items()}
m.profile_enable(None); m.profile_reset()
for _ in range(3): m.encode_text(toks)
torch.cuda.synchronize()
print({t: round(m.profile_read(t)[0] / 3, 3) for t in PROFILE_TAGS if m.profile_read(t)[1]})
m.profile_disable()
# config 3: 4096 tiles x 64 prompts, sim matrix + argmax
tiles = torch.randn(4096, 3, 224, 224, device="cuda").to(torch.bfloat16)
for _ in range(2):
    sim, lab = m.similarity(m.encode_image(tiles), m.encode_text(toks), mode="argmax")
torch.cuda.synchronize(); t0 = time.perf_counter()
sim, lab = m.similarity(m.encode_image(tiles), m.encode_text(toks), mode="argmax")
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"config 3 (4096 tiles x 64 prompts -> sim [4096,64] + argmax): {dt*1e3:.1f} ms total, {4096/dt:.0f} tiles/s")

#!/usr/bin/env python3
"""Small-batch latency: wall time per call vs the sum of kernel times (how launch-bound the small cases are)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keep_amd import KEEPModel, PROFILE_TAGS
from keep_amd.config import KEEPShape
from keep_amd.synth import synth_prompts, synth_state_dict

sd = synth_state_dict(KEEPShape(), seed=0)
from keep_amd.synth import towers_of
m = KEEPModel(towers=towers_of(sd)); m.load_state_dict(sd); m.to("cuda:0")
for a in sys.argv[1:]:
    k, v = a.split("="); m.set_option(k, float(v))
def wall(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def kern(fn, n=5):
    m.profile_enable(None); m.profile_reset()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    tot = sum(m.profile_read(t)[0] for t in PROFILE_TAGS) / n
    cnt = sum(m.profile_read(t)[1] for t in PROFILE_TAGS) / n
    m.profile_disable(); return tot, cnt
for B in (1, 2, 4, 8, 16, 32, 64):
    x = torch.randn(B, 3, 224, 224, device="cuda").to(torch.bfloat16)
    f = lambda: m.encode_image(x)
    w = wall(f); k, c = kern(f)
    print(f"encode_image B={B:3d}: wall {w:7.3f} ms  kernels {k:7.3f} ms  ({c:.0f} profiled regions)", flush=True)
for P in (1, 4, 16, 64):
    toks = {k: v.cuda() for k, v in synth_prompts(P, 256, seed=1).items()}
    f = lambda: m.encode_text(toks)
    w = wall(f); k, c = kern(f)
    print(f"encode_text  P={P:3d} (T run {m.last_text_length}): wall {w:7.3f} ms  kernels {k:7.3f} ms  ({c:.0f} profiled regions)", flush=True)
# CPU issue time vs GPU time for the launch-bound cases (no host syncs inside the call)
m.check_token_ids = False; m.trim_padding = False
for P in (1, 16):
    toks = {k: v[:, :32].contiguous().cuda() for k, v in synth_prompts(P, 256, seed=1).items()}
    for k in toks: toks[k][:, :] = toks[k]
    toks["attention_mask"][:, :8] = 1
    for _ in range(5): m.encode_text(toks)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): m.encode_text(toks)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"encode_text P={P} T=32, 50 calls back to back: CPU issue {1e3*(t1-t0)/50:.3f} ms/call, until GPU done {1e3*(t2-t0)/50:.3f} ms/call", flush=True)
x = torch.randn(1, 3, 224, 224, device="cuda").to(torch.bfloat16)
for _ in range(5): m.encode_image(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): m.encode_image(x)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"encode_image B=1, 50 calls back to back: CPU issue {1e3*(t1-t0)/50:.3f} ms/call, until GPU done {1e3*(t2-t0)/50:.3f} ms/call", flush=True)

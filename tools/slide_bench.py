#!/usr/bin/env python3
"""Slide-level steps at BASELINE config 4/5 size (N = 100 000 tiles, K = 1782 prompt sets x C = 4 classes) on ready features."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from keep_amd import KEEPModel, wsi
m = wsi._engine(KEEPModel())
N, K, C, D = 100_000, 1782, 4, 768
g = torch.Generator(device="cuda").manual_seed(0)
feats = torch.randn(N, D, device="cuda", generator=g)
cls = [torch.nn.functional.normalize(torch.randn(D, C, device="cuda", generator=g), dim=0) for _ in range(K)]
gx = 400
cells = torch.randperm(gx * 250, device="cuda")[:N]
coords = torch.stack([(cells % gx) * 256, (cells // gx) * 256], 1).cpu().numpy()
def t(name, f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{name:58s} {dt*1e3:9.2f} ms", flush=True); return r
fn = t("normalize features [100k,768]", lambda: wsi._normalized(m, feats))
bank = torch.stack([c.t() for c in cls]).reshape(K * C, D).contiguous()
for mode, name in ((1, "fused, fp16 + MX-fp4 corrections, top-2 in registers"), (2, "fused, three fp16 passes"), (0, "unfused: fp32 MFMA GEMM, logits through HBM")):
    m.set_option("fused_screening", mode)
    sc_m = t(f"prompt_scores [100k,768]x[768,{K*C}] (1.09 TFLOP): {name}", lambda: wsi.prompt_scores(fn, cls, pre_normalized=True, model=m), 3)
    if mode == 1: sc = sc_m
    else: print(f"    max |score - fused score| = {(sc_m - sc).abs().max().item():.2e}")
m.set_option("fused_screening", 1)
ens = t("zero_shot_prompt_select (scores + sort + merge top 50)", lambda: wsi.zero_shot_prompt_select(cls, feats, 50, "cuda:0", model=m), 3)
pr = t("probabilities softmax(10 cos) [100k,4]", lambda: wsi._probs(m, ens, feats))
t("refine (coordinate hash + 2x2 neighbour mean)", lambda: wsi.refine(pr, coords, 256, True, model=m))
t("zero_shot_subtyping end to end (given classifier)", lambda: wsi.zero_shot_subtyping(ens, feats, coords, 256, True, model=m))
t("zero_shot_detection end to end", lambda: wsi.zero_shot_detection(ens[:, :2].contiguous(), feats, coords, 256, False, model=m))
t("zero_shot_segment_probs end to end (dict of 100k entries)", lambda: wsi.zero_shot_segment_probs(ens[:, :2].contiguous(), feats, coords, 224, True, model=m), 2)

#!/usr/bin/env python3
"""Tile x prompt similarity / probability-map throughput against its HBM roofline (bytes = s*(768 N + 768 P + N P))."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keep_amd import KEEPModel
m = KEEPModel(); m.to("cuda:0") if hasattr(m, "to") else None
N = 100_000
img = torch.nn.functional.normalize(torch.randn(N, 768, device="cuda"), dim=-1)
for P, mode, scale, osz in ((2, "softmax_f16", 10.0, 2), (2, "softmax", 10.0, 4), (4, "softmax", 10.0, 4), (64, "argmax", 1.0, 4), (64, "raw", 1.0, 4)):
    txt = torch.nn.functional.normalize(torch.randn(P, 768, device="cuda"), dim=-1)
    for _ in range(3): m.similarity(img, txt, scale=scale, mode=mode)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): m.similarity(img, txt, scale=scale, mode=mode)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    byts = 4 * 768 * (N + P) + osz * N * P
    print(f"N={N} P={P:3d} {mode:12s}: {dt*1e6:8.1f} us  {byts/dt/1e9:7.0f} GB/s algorithmic ({byts/1e6:.0f} MB)  {2*N*P*768/dt/1e12:6.2f} TFLOP/s", flush=True)

#!/usr/bin/env python3
"""Time the image-tower attention kernel alone (197 tokens, 16 heads, B images per launch) on the library KEEP_HIP_LIB points at.
    KEEP_HIP_LIB=keep_amd/libkeep_hip_x.so python tools/attn_time.py [--batch 128]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keep_amd.ops import Ops

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=128); a = ap.parse_args()
o = Ops("cuda:0")
g = torch.Generator().manual_seed(0)
for B in (a.batch, 2 * a.batch):
    qkv = (torch.randn(B * 197, 3 * 16 * 64, generator=g) * 1.5).cuda()
    for _ in range(3): out = o.attention(qkv, B, 197, 16, None, False)
    torch.cuda.synchronize()
    # Ops.attention stages its input every call: time a batch of calls and subtract nothing -- compare LIBRARIES, not absolute numbers
    t0 = time.perf_counter()
    for _ in range(20): out = o.attention(qkv, B, 197, 16, None, False)
    torch.cuda.synchronize()
    print(f"{os.environ.get('KEEP_HIP_LIB', 'default')} B={B}: {(time.perf_counter() - t0) / 20 * 1e6:.1f} us per call (incl. staging); checksum {float(out.float().abs().sum()):.4f}")

#!/usr/bin/env python3
"""Per-workgroup phase timeline of the attention kernel (shader clocks): staging (K by LDS-DMA, V^T through registers)
vs compute (S^T, softmax, O^T) for the ViT-L shape, 128 tiles x 16 heads.

Needs a diagnostics build of the library (the timing / ablation hooks are compiled out of the product .so):
    KEEP_BUILD_DEFINES="-DKEEP_DIAGNOSTICS" KEEP_BUILD_OUT=libkeep_hip_diag.so python -m keep_amd.build
    KEEP_HIP_LIB=$PWD/keep_amd/libkeep_hip_diag.so python tools/attn_timeline.py
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from keep_amd.ops import Ops
ops = Ops("cuda:0")
ops.set_option("gemm_dbg", 1)
for a in sys.argv[1:]:
    k, v = a.split("="); ops.set_option(k, float(v))
B, T, H = 128, 197, 16
qkv = torch.randn(B * T, 3 * H * 64, device="cuda") * 0.5
for _ in range(3): ops.attention(qkv, B, T, H)
t = ops.debug_timeline(B * H).astype(np.float64)
t0 = t[:, 0].min()
stage, comp, tot = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 2] - t[:, 0]
print(f"workgroups {B*H}: kernel span {t[:, 2].max() - t0:.0f} clk; per workgroup staging {stage.mean():.0f} (p10/50/90 {np.percentile(stage,[10,50,90]).astype(int)}) "
      f"compute {comp.mean():.0f} (p10/50/90 {np.percentile(comp,[10,50,90]).astype(int)}) total {tot.mean():.0f}")
print(f"span / (workgroups / 512 resident) = {(t[:, 2].max() - t0) / (B * H / 512):.0f} clk per round of 2 workgroups per CU")

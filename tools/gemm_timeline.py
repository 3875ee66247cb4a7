#!/usr/bin/env python3
"""Per-workgroup phase timeline of the 256x256 GEMM (shader clocks) for the ViT-L shapes.

Needs a diagnostics build of the library (the timing / ablation hooks are compiled out of the product .so):
    KEEP_BUILD_DEFINES="-DKEEP_DIAGNOSTICS" KEEP_BUILD_OUT=libkeep_hip_diag.so python -m keep_amd.build
    KEEP_HIP_LIB=$PWD/keep_amd/libkeep_hip_diag.so python tools/gemm_timeline.py [name=value engine options ...]

    gemm_persistent=0|1      one tile per workgroup / the persistent walk (stamps then describe a workgroup's last tile)
    gemm_ablate=$((D*256))   workgroups start (slot & 3) * D * 1024 cycles late: four phase groups inside every XCD (how long is an epilogue when the
                             chip is not in its epilogue all at once?  profiles/r05_epilogue_dephasing.txt)
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from keep_amd.ops import Ops, EPI_F16, EPI_GELU_F16, EPI_RESID_LS

ops = Ops("cuda:0")
ops.set_option("gemm_impl", 256)
ops.set_option("gemm_dbg", 1)
COMP = False
M = 50432
for a in sys.argv[1:]:
    k, v = a.split("=")
    if k == "M":                 # rows: 50 432 = 256 tiles x 197 tokens (default); M=4096 leaves most CUs idle (is an epilogue bound by the whole chip's HBM rate?)
        M = int(v)
    elif k == "comp":              # comp=1: compensated launches; stamp 1 then marks the end of the fp16 phase, "loop" below is the MX-fp4 phase
        COMP = bool(int(v))
    else:
        ops.set_option(k, float(v))
for name, N, K, epi in (("qkv", 3072, 1024, EPI_F16), ("proj", 1024, 1024, EPI_RESID_LS), ("fc1", 4096, 1024, EPI_GELU_F16), ("fc2", 1024, 4096, EPI_RESID_LS)):
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.03; b = torch.zeros(N, device="cuda")
    ls = torch.ones(N, device="cuda"); r = torch.randn(M, N, device="cuda") if epi == EPI_RESID_LS else None
    for _ in range(2):
        ops.linear(a, w, b, epi, 2 if COMP else False, ls=ls, resid=r)
    nb = (M // 256) * (N // 256)
    t = ops.debug_timeline(nb).astype(np.float64)
    t = t[t[:, 3] > 0]          # a persistent launch (gemm_persistent=1) stamps one row per workgroup, describing its LAST tile
    t0 = t[:, 0].min()
    dur = t[:, 3].max() - t0
    pro, loop, epi_t, tot = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 3] - t[:, 0]
    first = np.sort(t[:, 0] - t0)
    if COMP:
        print(f"{name:5s} blocks={nb} kernel span={dur:9.0f} clk | per block: prologue + fp16 phase {pro.mean():8.0f}  fp4 phase {loop.mean():8.0f} "
              f"({loop.mean() / (K // 64):6.0f}/chunk of K=64 incl. its prologue)  epilogue {epi_t.mean():7.0f}  total {tot.mean():8.0f}")
        continue
    print(f"{name:5s} blocks={nb} kernel span={dur:9.0f} clk | per block: prologue {pro.mean():7.0f} loop {loop.mean():8.0f} "
          f"({loop.mean() / (K // 32):6.0f}/step) epilogue {epi_t.mean():7.0f} total {tot.mean():8.0f} | "
          f"start times pct[10,50,90]={np.percentile(first, [10, 50, 90]).astype(int)} ; "
          f"epi pct[10,50,90]={np.percentile(epi_t, [10, 50, 90]).astype(int)}")

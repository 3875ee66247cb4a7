"""Calibration only (not part of the product path): what the vendor GEMM reaches on this box for the ViT-L shapes,
to judge how far the hand-written kernel is from a practical ceiling.  torch.matmul -> hipBLASLt/rocBLAS."""
import torch, time
M = 197 * 256
for name, N, K in [("qkv", 3072, 1024), ("proj", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096)]:
    for dt in (torch.float16, torch.bfloat16):
        a = torch.randn(M, K, device="cuda", dtype=dt); w = torch.randn(N, K, device="cuda", dtype=dt)
        for _ in range(5): torch.matmul(a, w.t())
        torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True)
        e0.record()
        for _ in range(20): torch.matmul(a, w.t())
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{name:5s} {str(dt):15s} M={M} N={N} K={K}: {ms:.3f} ms  {2*M*N*K/ms/1e9:.0f} TFLOP/s", flush=True)

import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from keep_amd.ops import Ops
from test_ops_gpu import attn_ref, rand
ops = Ops("cuda:0")
for (B, T, heads) in ((1, 17, 1), (1, 100, 2), (2, 64, 12), (3, 197, 16)):
    qkv = rand(B * T, 3 * heads * 64, seed=20, std=1.5)
    for split in (False, True):
        out = ops.attention(qkv, B, T, heads, None, split).cpu().double()
        ref = attn_ref(qkv, B, T, heads, None, not split)
        d = (out - ref).abs()
        print(B, T, heads, split, float(d.max()), "rows with err>1e-4:", int((d.max(1).values > 1e-4).sum()), "of", d.shape[0], "finite", bool(torch.isfinite(out).all()))

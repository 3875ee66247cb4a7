#!/usr/bin/env python3
"""CPU emulation (torch fp32 + fp16 operand rounding): WHERE on the attention side (and the MLP) does the single-pass fp16 error of the image
tower come from, per tile family -- and what would a CLS-row-only treatment of the attention side leave?

Every variant runs ALL sites plain (fp16-rounded operands) except the named class, which is exact:
   none        everything plain (the yardstick)
   qkv_gemm    the qkv GEMM's operands exact
   qkv_store   q / k / v not rounded to fp16 before QK^T / PV
   proj_gemm   the proj GEMM's operands (the attention output's storage rounding and W_proj's) exact
   attn_all    the three above
   mlp_all     fc1 / fc2 operands exact
   attn_cls    attention side plain for every row, then the CLS row again exactly: its q (exact LN-1 row x W_qkv), attention over the plain K / V
               (its own k / v exact), proj exact -- the analogue of KEEP_MLP_CLS
   proj_cls    the CLS row's proj again exactly, on the unrounded attention output of the plain q / k / v (the cheap half of attn_cls)
   mlp_cls     the CLS row's MLP again exactly (KEEP_MLP_CLS)
   both_cls    attn_cls + mlp_cls
    python tools/attn_site_study.py [--tiles 8] [--depth 24]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import keep_oracle as O                                     # noqa: E402
from keep_amd.config import KEEPShape                                   # noqa: E402
from keep_amd.synth import normalise_u8, synth_state_dict, synth_tile_family, synth_tiles   # noqa: E402

r16 = lambda x: x.to(torch.float16).to(torch.float32)


def lin(x, w, b, exact):
    return x @ w.t() + b if exact else r16(x) @ r16(w).t() + b


def features(sd, x, depth, exact=(), cls_attn=False, cls_mlp=False, cls_proj=False):
    heads, eps = 16, 1e-6
    B = x.shape[0]
    wpe = sd["visual.patch_embed.proj.weight"]
    D = wpe.shape[0]
    p = O.patchify(x, 16) @ wpe.reshape(D, -1).t() + sd["visual.patch_embed.proj.bias"]
    t = torch.cat([sd["visual.cls_token"].expand(B, -1, -1), p], 1) + sd["visual.pos_embed"]
    N, hd = t.shape[1], D // heads
    for i in range(depth):
        bp = f"visual.blocks.{i}."
        g = lambda k: sd[bp + k]
        h = O.layer_norm(t, g("norm1.weight"), g("norm1.bias"), eps)
        qkv = lin(h, g("attn.qkv.weight"), g("attn.qkv.bias"), "qkv_gemm" in exact)
        qkv_s = qkv if "qkv_store" in exact else r16(qkv)
        q, k, v = qkv_s.reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
        a = O._sdpa(q, k, v, None).transpose(1, 2).reshape(B, N, D)
        y = lin(a, g("attn.proj.weight"), g("attn.proj.bias"), "proj_gemm" in exact)
        t_new = t + g("ls1.gamma") * y
        if cls_attn:
            qkv0 = h[:, :1] @ g("attn.qkv.weight").t() + g("attn.qkv.bias")                      # [B,1,3D] exact
            q0, k0, v0 = qkv0.reshape(B, 1, 3, heads, hd).permute(2, 0, 3, 1, 4)
            k2, v2 = k.clone(), v.clone()
            k2[:, :, :1], v2[:, :, :1] = k0, v0
            a0 = O._sdpa(q0, k2, v2, None).transpose(1, 2).reshape(B, 1, D)
            t_new[:, :1] = t[:, :1] + g("ls1.gamma") * (a0 @ g("attn.proj.weight").t() + g("attn.proj.bias"))
        if cls_proj:          # the CLS row's proj again, exactly, on the UNROUNDED attention output of the plain q / k / v (what an fp32 accumulator holds)
            t_new[:, :1] = t[:, :1] + g("ls1.gamma") * (a[:, :1] @ g("attn.proj.weight").t() + g("attn.proj.bias"))
        t = t_new
        h = O.layer_norm(t, g("norm2.weight"), g("norm2.bias"), eps)
        ex = "mlp" in exact
        m = O.gelu_erf(lin(h, g("mlp.fc1.weight"), g("mlp.fc1.bias"), ex))
        t_new = t + g("ls2.gamma") * lin(m, g("mlp.fc2.weight"), g("mlp.fc2.bias"), ex)
        if cls_mlp:
            m0 = O.gelu_erf(h[:, :1] @ g("mlp.fc1.weight").t() + g("mlp.fc1.bias"))
            t_new[:, :1] = t[:, :1] + g("ls2.gamma") * (m0 @ g("mlp.fc2.weight").t() + g("mlp.fc2.bias"))
        t = t_new
    f = O.layer_norm(t, sd["visual.norm.weight"], sd["visual.norm.bias"], eps)[:, 0]
    return O.l2_normalize(O.visual_head(sd, f))


VARIANTS = {"none": dict(), "qkv_gemm": dict(exact=("qkv_gemm",)), "qkv_store": dict(exact=("qkv_store",)), "proj_gemm": dict(exact=("proj_gemm",)),
            "attn_all": dict(exact=("qkv_gemm", "qkv_store", "proj_gemm")), "mlp_all": dict(exact=("mlp",)),
            "proj_cls": dict(cls_proj=True), "proj+mlp_cls": dict(cls_proj=True, cls_mlp=True), "attn_cls": dict(cls_attn=True), "mlp_cls": dict(cls_mlp=True), "both_cls": dict(cls_attn=True, cls_mlp=True)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=8)
    ap.add_argument("--depth", type=int, default=24)
    ap.add_argument("--variants", nargs="*", default=None)
    ap.add_argument("--families", nargs="*", default=["gaussian", "he_crops", "stain_field", "background", "half"])
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 8)
    sd = synth_state_dict(KEEPShape(), seed=0, text=False)
    if args.variants:
        for k in list(VARIANTS):
            if k not in args.variants:
                del VARIANTS[k]
    print(f"{'family':12s} " + " ".join(f"{v:>12s}" for v in VARIANTS) + "    (isotropic rms cosine error x 1e5)")
    with torch.no_grad():
        for fam in args.families:
            x = synth_tiles(args.tiles, seed=5) if fam == "gaussian" else normalise_u8(synth_tile_family(fam, 0, args.tiles, "cpu", seed=7001))
            exact_all = ("qkv_gemm", "qkv_store", "proj_gemm", "mlp")
            ref = features(sd, x, args.depth, exact=exact_all)
            row = [float((features(sd, x, args.depth, **kw) - ref).pow(2).sum(1).mean().div(768).sqrt()) * 1e5 for kw in VARIANTS.values()]
            print(f"{fam:12s} " + " ".join(f"{v:12.3f}" for v in row), flush=True)


if __name__ == "__main__":
    main()

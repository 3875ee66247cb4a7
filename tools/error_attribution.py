#!/usr/bin/env python3
"""Which fp16 operand roundings make up the fp16-mode cosine error?  CPU experiment with the oracle's arithmetic:
full-depth ViT-L on seeded synthetic weights, fp32 everywhere except ONE class of operands rounded to fp16, 8 tiles x 64
random unit text vectors (512 cosines) against the unrounded run.  Variances add, so the rms^2 shares show where a second
MFMA pass would buy the most."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keep_amd.config import KEEPShape
from keep_amd.synth import synth_state_dict, synth_tiles
from oracle import keep_oracle as O

torch.set_num_threads(os.cpu_count())
sd = synth_state_dict(KEEPShape(), seed=0, text=False)
x = synth_tiles(8, seed=100)
bank = torch.nn.functional.normalize(torch.randn(64, 768, generator=torch.Generator().manual_seed(3)), dim=-1)
r16 = lambda t: t.to(torch.float16).to(torch.float32)

def forward(sites):
    """sites: set of strings like 'qkv.A', 'qkv.W', 'attn', 'proj.A', ... ; 'all'."""
    on = lambda s: "all" in sites or s in sites
    def lin(name, a, w, b):
        if on(name + ".A"): a = r16(a)
        if on(name + ".W"): w = r16(w)
        return a @ w.t() + b
    p = "visual."
    wpe = sd[p + "patch_embed.proj.weight"]
    t = O.patchify(x, 16) @ wpe.reshape(1024, -1).t() + sd[p + "patch_embed.proj.bias"]      # patch embed is always strict in the engine
    t = torch.cat([sd[p + "cls_token"].expand(x.shape[0], -1, -1), t], 1) + sd[p + "pos_embed"]
    for i in range(24):
        bp = f"{p}blocks.{i}."
        h = O.layer_norm(t, sd[bp + "norm1.weight"], sd[bp + "norm1.bias"], 1e-6)
        qkv = lin("qkv", h, sd[bp + "attn.qkv.weight"], sd[bp + "attn.qkv.bias"])
        if on("qkv.out"): qkv = r16(qkv)                      # q, k, v are stored as fp16 (attention operands)
        qkv = qkv.reshape(x.shape[0], 197, 3, 16, 64).permute(2, 0, 3, 1, 4)
        s = (qkv[0] @ qkv[1].transpose(-1, -2)) * 0.125
        pr = torch.softmax(s, -1)
        if on("attn.P"): pr = r16(pr)
        a = (pr @ qkv[2]).transpose(1, 2).reshape(x.shape[0], 197, 1024)
        t = t + sd[bp + "ls1.gamma"] * lin("proj", a, sd[bp + "attn.proj.weight"], sd[bp + "attn.proj.bias"])
        h = O.layer_norm(t, sd[bp + "norm2.weight"], sd[bp + "norm2.bias"], 1e-6)
        m = O.gelu_erf(lin("fc1", h, sd[bp + "mlp.fc1.weight"], sd[bp + "mlp.fc1.bias"]))
        t = t + sd[bp + "ls2.gamma"] * lin("fc2", m, sd[bp + "mlp.fc2.weight"], sd[bp + "mlp.fc2.bias"])
    f = O.layer_norm(t, sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-6)[:, 0]
    return O.l2_normalize(O.visual_head(sd, f))

with torch.no_grad():
    ref = forward(set()) @ bank.t()
    rows = []
    for sites in (["all", "qkv.out", "attn.P"], ["qkv.A"], ["qkv.W"], ["qkv.out"], ["attn.P"], ["proj.A"], ["proj.W"], ["fc1.A"], ["fc1.W"], ["fc2.A"], ["fc2.W"]):
        d = forward(set(sites)) @ bank.t() - ref
        rows.append((sites[0] if len(sites) == 1 else "everything", d.abs().max().item(), d.pow(2).mean().sqrt().item()))
        print(f"{rows[-1][0]:12s} max|dcos| {rows[-1][1]:.2e}  rms {rows[-1][2]:.2e}", flush=True)
tot = sum(r[2] ** 2 for r in rows[1:])
print("shares of the summed variance:", {r[0]: round(r[2] ** 2 / tot, 3) for r in rows[1:]}, f"; sqrt(sum) = {math.sqrt(tot):.2e} vs everything {rows[0][2]:.2e}")

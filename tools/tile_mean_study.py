#!/usr/bin/env python3
"""CPU emulation (torch fp32 + fp16 operand rounding, the oracle's error model): how much of the single-pass fp16 GEMM error of the image tower is
the ROW-INDEPENDENT part W_lo @ (mean input row), per tile family -- and what removing it per TILE (mean over the tile's 197 rows) instead of once
for a global probe mean (round 5's `keep_calibrate_bias`) would buy.

variants: plain        every linear GEMM of the blocks on fp16-rounded operands (attention operands rounded too)
          static_g     + W_lo @ mean row of N(0,1) probe tiles folded into the bias (round 5)
          tile_mean    + W_lo @ (this tile's mean input row) added to every row of the tile
          tile_centred (ideal) W @ mean row in fp32 + fp16 GEMM on the centred rows
    python tools/tile_mean_study.py [--tiles 8] [--depth 24]
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

H = torch.float16


def r16(x):
    return x.to(H).to(torch.float32)


class Tower:
    def __init__(self, sd, depth):
        self.sd, self.depth = sd, depth
        self.static = {}          # site key -> mean input row of the static probe

    def linear(self, x, key, variant, record=None):
        w, b = self.sd[key + ".weight"], self.sd[key + ".bias"]
        if variant == "exact":
            if record is not None:
                record[key] = x.reshape(-1, x.shape[-1]).mean(0)
            return x @ w.t() + b
        wh = r16(w)
        wl = w - wh
        if variant == "tile_centred":
            m = x.mean(dim=1, keepdim=True)
            return r16(x - m) @ wh.t() + m @ w.t() + b
        y = r16(x) @ wh.t() + b
        if variant == "static_g":
            y = y + self.static[key] @ wl.t()
        elif variant == "tile_mean":
            y = y + x.mean(dim=1, keepdim=True) @ wl.t()
        return y

    def features(self, x, variant, record=None):
        sd, heads, eps = self.sd, 16, 1e-6
        B = x.shape[0]
        wpe = sd["visual.patch_embed.proj.weight"]
        D = wpe.shape[0]
        p = O.patchify(x, 16) @ wpe.reshape(D, -1).t() + sd["visual.patch_embed.proj.bias"]
        t = torch.cat([sd["visual.cls_token"].expand(B, -1, -1), p], 1) + sd["visual.pos_embed"]
        N, hd = t.shape[1], D // heads
        for i in range(self.depth):
            bp = f"visual.blocks.{i}."
            h = O.layer_norm(t, sd[bp + "norm1.weight"], sd[bp + "norm1.bias"], eps)
            qkv = self.linear(h, bp + "attn.qkv", variant, record).reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
            q, k, v = qkv[0], qkv[1], qkv[2]
            if variant != "exact":
                q, k, v = r16(q), r16(k), r16(v)
            a = O._sdpa(q, k, v, None).transpose(1, 2).reshape(B, N, D)
            t = t + sd[bp + "ls1.gamma"] * self.linear(a, bp + "attn.proj", variant, record)
            h = O.layer_norm(t, sd[bp + "norm2.weight"], sd[bp + "norm2.bias"], eps)
            m = O.gelu_erf(self.linear(h, bp + "mlp.fc1", variant, record))
            t = t + sd[bp + "ls2.gamma"] * self.linear(m, bp + "mlp.fc2", variant, record)
        f = O.layer_norm(t, sd["visual.norm.weight"], sd["visual.norm.bias"], eps)[:, 0]
        return O.l2_normalize(O.visual_head(sd, f))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=8)
    ap.add_argument("--depth", type=int, default=24)
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 8)
    sd = synth_state_dict(KEEPShape(), seed=0, text=False)
    tw = Tower(sd, args.depth)
    with torch.no_grad():
        rec = {}
        tw.features(synth_tiles(args.tiles, seed=77), "exact", rec)
        tw.static = rec
        print(f"{'family':12s} " + " ".join(f"{v:>13s}" for v in ("plain", "static_g", "tile_mean", "tile_centred")) + "   (isotropic rms cosine error; systematic share of plain)")
        for fam in ("gaussian", "he_crops", "stain_field", "background", "half"):
            x = synth_tiles(args.tiles, seed=5) if fam == "gaussian" else normalise_u8(synth_tile_family(fam, 0, args.tiles, "cpu", seed=7001))
            ref = tw.features(x, "exact")
            row = []
            for variant in ("plain", "static_g", "tile_mean", "tile_centred"):
                e = tw.features(x, variant) - ref
                row.append(float(e.pow(2).sum(1).mean().div(768).sqrt()))
                if variant == "plain":
                    syst = float(e.mean(0).norm() ** 2 / e.pow(2).sum(1).mean())
            print(f"{fam:12s} " + " ".join(f"{v:13.3e}" for v in row) + f"   {syst:.2f}", flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""CPU emulation (torch fp32 + fp16 operand rounding): the text tower with single fp16 passes for every row and the [CLS] row of every prompt redone exactly
(its q / k / v, its attention over the plain keys, attention-output dense, FFN) -- the analogue of the image tower's CLS-row chain, which the round-5 review asked
about for BERT (the pooled text feature is tanh(W h[:, 0]), keep_inference.py:61).  Reports the cosine error the TEXT side alone would add to every cosine of a slide
against random unit image features (isotropic rms = |text feature error| / sqrt 768).
    python tools/text_cls_study.py [--prompts 64]
"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import keep_oracle as O                                     # noqa: E402
from keep_amd.config import KEEPShape                                   # noqa: E402
from keep_amd.synth import synth_prompts, synth_state_dict              # noqa: E402

r16 = lambda x: x.to(torch.float16).to(torch.float32)


def lin(x, w, b, exact):
    return x @ w.t() + b if exact else r16(x) @ r16(w).t() + b


def pooled(sd, toks, mode, heads=12, eps=1e-12):
    """mode: 'exact' | 'plain' | 'cls' (plain for every row + the [CLS] row exactly)."""
    ids, mask = toks["input_ids"], toks["attention_mask"]
    P, T = ids.shape
    p = "text."
    e = sd[p + "embeddings.word_embeddings.weight"][ids] + sd[p + "embeddings.token_type_embeddings.weight"][torch.zeros_like(ids)] + sd[p + "embeddings.position_embeddings.weight"][:T][None]
    h = O.layer_norm(e, sd[p + "embeddings.LayerNorm.weight"], sd[p + "embeddings.LayerNorm.bias"], eps)
    H = h.shape[-1]; hd = H // heads
    bias = (1.0 - mask[:, None, None, :].float()) * torch.finfo(torch.float32).min
    ex = mode == "exact"
    for i in range(O.count_bert_layers(sd)):
        lp = f"{p}encoder.layer.{i}."
        g = lambda k: sd[lp + k]
        hs = lambda z: z.reshape(P, -1, heads, hd).transpose(1, 2)
        q = lin(h, g("attention.self.query.weight"), g("attention.self.query.bias"), ex)
        k = lin(h, g("attention.self.key.weight"), g("attention.self.key.bias"), ex)
        v = lin(h, g("attention.self.value.weight"), g("attention.self.value.bias"), ex)
        if not ex:
            q, k, v = r16(q), r16(k), r16(v)
        a = O._sdpa(hs(q), hs(k), hs(v), bias).transpose(1, 2).reshape(P, T, H)
        o = lin(a, g("attention.output.dense.weight"), g("attention.output.dense.bias"), ex)
        h1 = O.layer_norm(h + o, g("attention.output.LayerNorm.weight"), g("attention.output.LayerNorm.bias"), eps)
        if mode == "cls":
            h0 = h[:, :1]
            q0 = h0 @ g("attention.self.query.weight").t() + g("attention.self.query.bias")
            k2, v2 = k.clone(), v.clone()
            k2[:, :1] = h0 @ g("attention.self.key.weight").t() + g("attention.self.key.bias")
            v2[:, :1] = h0 @ g("attention.self.value.weight").t() + g("attention.self.value.bias")
            a0 = O._sdpa(hs(q0), hs(k2), hs(v2), bias).transpose(1, 2).reshape(P, 1, H)
            o0 = a0 @ g("attention.output.dense.weight").t() + g("attention.output.dense.bias")
            h1[:, :1] = O.layer_norm(h0 + o0, g("attention.output.LayerNorm.weight"), g("attention.output.LayerNorm.bias"), eps)
        m = O.gelu_erf(lin(h1, g("intermediate.dense.weight"), g("intermediate.dense.bias"), ex))
        o = lin(m, g("output.dense.weight"), g("output.dense.bias"), ex)
        h2 = O.layer_norm(h1 + o, g("output.LayerNorm.weight"), g("output.LayerNorm.bias"), eps)
        if mode == "cls":
            x0 = h1[:, :1]
            m0 = O.gelu_erf(x0 @ g("intermediate.dense.weight").t() + g("intermediate.dense.bias"))
            h2[:, :1] = O.layer_norm(x0 + m0 @ g("output.dense.weight").t() + g("output.dense.bias"), g("output.LayerNorm.weight"), g("output.LayerNorm.bias"), eps)
        h = h2
    return O.l2_normalize(torch.tanh(h[:, 0] @ sd[p + "pooler.dense.weight"].t() + sd[p + "pooler.dense.bias"]))


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--prompts", type=int, default=64); a = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 8)
    sd = synth_state_dict(KEEPShape(), seed=0, vision=False)
    toks = synth_prompts(a.prompts, 32, seed=1)               # (trimmed to the longest valid length: same result as the padded call)
    with torch.no_grad():
        ref = pooled(sd, toks, "exact")
        for mode in ("plain", "cls"):
            e = pooled(sd, toks, mode) - ref
            print(f"text tower, {mode:5s}: text-side isotropic rms cosine error {float(e.pow(2).sum(1).mean().div(768).sqrt()):.3e}   worst prompt {float(e.pow(2).sum(1).max().div(768).sqrt()):.3e}")
    print("(plain on the GPU: 4.0e-5 rms against 512 probe tiles' image features, profiles/r06_text_encode_rates.txt; the image side's whole budget is 1.57e-5)")


if __name__ == "__main__":
    main()

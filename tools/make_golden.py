#!/usr/bin/env python3
"""Pin oracle/keep_oracle.py against real implementations and write tests/golden/*.npz.

Runs ONLY in the build container (needs `transformers` and /root/reference):

  text tower   <- transformers.BertModel (the class the reference instantiates,
                  quick_start/keep_inference.py:49-50)
  image tower  <- transformers.Dinov2Model configured as ViT-L/16 + LayerScale
                  (independent implementation of timm vit_large_patch16_224's
                  block arithmetic; timm is not installed here)
  WSI logic    <- /root/reference/WSI_evaluation/{utils,subtyping_utils,
                  detection_utils,segment_utils}.py imported as-is (h5py and
                  openslide stubbed: they are import-time only for these functions)

Each section asserts the oracle agrees, then stores inputs that cannot be
regenerated from a seed plus the expected outputs.  Weights and tiles are NOT
stored (1.6 GB): they are regenerated from keep_amd.synth seeds; a checksum of
the seeded tensors is stored so generator drift is detected.

    python tools/make_golden.py            # all sections
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from keep_amd.config import KEEPShape, small_shape          # noqa: E402
from keep_amd.synth import synth_prompts, synth_state_dict, synth_tiles   # noqa: E402
from oracle import keep_oracle as O                          # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"


def checksum(t: torch.Tensor) -> float:
    return float(t.double().abs().sum())


# --------------------------------------------------------------------------
def hf_dinov2_from_sd(sd, depth):
    from transformers import Dinov2Config, Dinov2Model
    cfg = Dinov2Config(hidden_size=1024, num_hidden_layers=depth, num_attention_heads=16, mlp_ratio=4,
                       image_size=224, patch_size=16, layerscale_value=1.0, layer_norm_eps=1e-6,
                       qkv_bias=True, use_swiglu_ffn=False, hidden_act="gelu",
                       attn_implementation="eager")
    m = Dinov2Model(cfg).eval()
    new = {"embeddings.cls_token": sd["visual.cls_token"],
           "embeddings.mask_token": torch.zeros(1, 1024),
           "embeddings.position_embeddings": sd["visual.pos_embed"],
           "embeddings.patch_embeddings.projection.weight": sd["visual.patch_embed.proj.weight"],
           "embeddings.patch_embeddings.projection.bias": sd["visual.patch_embed.proj.bias"],
           "layernorm.weight": sd["visual.norm.weight"], "layernorm.bias": sd["visual.norm.bias"]}
    for i in range(depth):
        s, d = f"visual.blocks.{i}.", f"encoder.layer.{i}."
        w, b = sd[s + "attn.qkv.weight"], sd[s + "attn.qkv.bias"]
        for j, name in enumerate(("query", "key", "value")):
            new[d + f"attention.attention.{name}.weight"] = w[j * 1024:(j + 1) * 1024]
            new[d + f"attention.attention.{name}.bias"] = b[j * 1024:(j + 1) * 1024]
        new[d + "attention.output.dense.weight"] = sd[s + "attn.proj.weight"]
        new[d + "attention.output.dense.bias"] = sd[s + "attn.proj.bias"]
        new[d + "layer_scale1.lambda1"] = sd[s + "ls1.gamma"]
        new[d + "layer_scale2.lambda1"] = sd[s + "ls2.gamma"]
        for n in ("norm1", "norm2"):
            new[d + n + ".weight"] = sd[s + n + ".weight"]
            new[d + n + ".bias"] = sd[s + n + ".bias"]
        for n in ("fc1", "fc2"):
            new[d + f"mlp.{n}.weight"] = sd[s + f"mlp.{n}.weight"]
            new[d + f"mlp.{n}.bias"] = sd[s + f"mlp.{n}.bias"]
    missing, unexpected = m.load_state_dict(new, strict=True), None
    return m


def golden_vit(depth: int, batch: int, seed: int):
    shape = small_shape(vit_depth=depth) if depth != 24 else KEEPShape()
    sd = synth_state_dict(shape, seed=seed, text=False)
    x = synth_tiles(batch, seed=seed + 100)
    with torch.no_grad():
        m = hf_dinov2_from_sd(sd, depth)
        out = m(pixel_values=x)
        cls_hf = out.pooler_output                      # LN'd CLS token  [B,1024]
        tok_hf = out.last_hidden_state
        head = torch.nn.Sequential(torch.nn.Linear(1024, 768), torch.nn.GELU(), torch.nn.Linear(768, 768))
        head[0].weight.copy_(sd["visual_head.0.weight"]); head[0].bias.copy_(sd["visual_head.0.bias"])
        head[2].weight.copy_(sd["visual_head.2.weight"]); head[2].bias.copy_(sd["visual_head.2.bias"])
        feat_hf = torch.nn.functional.normalize(head(cls_hf), dim=-1)    # keep_inference.py:56
        tok_or = O.vit_tokens(sd, x, depth)
        feat_or = O.encode_image(sd, x)
    d_tok = float((tok_or - tok_hf).abs().max())
    d_feat = float((feat_or - feat_hf).abs().max())
    print(f"[vit d{depth}] oracle vs Dinov2: max|dtok|={d_tok:.3e} max|dfeat|={d_feat:.3e}")
    assert d_tok < 5e-4 and d_feat < 2e-6, "oracle image tower disagrees with Dinov2-as-ViT-L"
    np.savez_compressed(os.path.join(GOLD, f"vit_d{depth}.npz"),
                        depth=depth, batch=batch, weight_seed=seed, tile_seed=seed + 100,
                        tiles_checksum=checksum(x), qkv0_checksum=checksum(sd["visual.blocks.0.attn.qkv.weight"]),
                        cls=cls_hf.numpy(), features=feat_hf.numpy(),
                        oracle_dtok=d_tok, oracle_dfeat=d_feat)


# --------------------------------------------------------------------------
def hf_bert_from_sd(sd, layers, impl):
    from transformers import BertConfig, BertModel
    cfg = BertConfig(vocab_size=30522, hidden_size=768, num_hidden_layers=layers, num_attention_heads=12,
                     intermediate_size=3072, max_position_embeddings=512, type_vocab_size=2,
                     layer_norm_eps=1e-12, hidden_act="gelu", attn_implementation=impl)
    m = BertModel(cfg).eval()
    new = {k[len("text."):]: v for k, v in sd.items() if k.startswith("text.")}
    res = m.load_state_dict(new, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert all("position_ids" in k or "token_type_ids" in k for k in res.missing_keys), res.missing_keys
    return m


def golden_bert(layers: int, n: int, seed: int):
    shape = small_shape(bert_layers=layers) if layers != 12 else KEEPShape()
    sd = synth_state_dict(shape, seed=seed, vision=False)
    toks = synth_prompts(n, 256, seed=seed + 200)
    # one fully-attended row and one short row to cover mask edge cases
    toks["attention_mask"][0, :] = 1
    toks["input_ids"][0] = torch.randint(4, 30522, (256,), generator=torch.Generator().manual_seed(seed + 201))
    toks["token_type_ids"][1, 5:9] = 1
    with torch.no_grad():
        outs = {}
        for impl in ("eager", "sdpa"):
            m = hf_bert_from_sd(sd, layers, impl)
            outs[impl] = m(**toks).pooler_output
        pooled_hf = outs["eager"]
        feat_hf = torch.nn.functional.normalize(pooled_hf, dim=-1)       # keep_inference.py:61
        feat_or = O.encode_text(sd, toks)
    d_impl = float((outs["eager"] - outs["sdpa"]).abs().max())
    d_feat = float((feat_or - feat_hf).abs().max())
    print(f"[bert l{layers}] eager vs sdpa {d_impl:.3e}; oracle vs BertModel max|dfeat|={d_feat:.3e}")
    assert d_feat < 2e-6 and d_impl < 2e-6
    np.savez_compressed(os.path.join(GOLD, f"bert_l{layers}.npz"),
                        layers=layers, weight_seed=seed,
                        input_ids=toks["input_ids"].numpy().astype(np.int32),
                        token_type_ids=toks["token_type_ids"].numpy().astype(np.int8),
                        attention_mask=toks["attention_mask"].numpy().astype(np.int8),
                        word_emb_checksum=checksum(sd["text.embeddings.word_embeddings.weight"]),
                        pooled=pooled_hf.numpy(), features=feat_hf.numpy(), oracle_dfeat=d_feat)


# --------------------------------------------------------------------------
def import_reference_wsi():
    sys.modules.setdefault("h5py", types.ModuleType("h5py"))
    sys.modules.setdefault("openslide", types.ModuleType("openslide"))
    sys.path.insert(0, os.path.join(REF, "WSI_evaluation"))
    import utils as r_utils                      # noqa
    import subtyping_utils as r_sub              # noqa
    import detection_utils as r_det              # noqa
    import segment_utils as r_seg                # noqa
    return r_utils, r_sub, r_det, r_seg


def golden_wsi(seed: int = 7):
    import contextlib, io
    r_utils, r_sub, r_det, r_seg = import_reference_wsi()
    g = torch.Generator().manual_seed(seed)
    N, K, D = 600, 40, 768
    # tile features with some class structure so that scores/labels are non-degenerate
    centers = torch.nn.functional.normalize(torch.randn(4, D, generator=g), dim=-1)
    lab = torch.randint(0, 4, (N,), generator=g)
    feats = centers[lab] * 1.5 + torch.randn(N, D, generator=g) * 0.8          # un-normalised, like h5 features
    gx, gy = 30, 25
    cells = torch.randperm(gx * gy, generator=g)[:N - 20]
    cells = torch.cat([cells, cells[:20]])                                     # 20 duplicate coordinates
    coords256 = torch.stack([(cells % gx) * 256, (cells // gx) * 256], dim=1).numpy()
    coords224 = torch.stack([(cells % gx) * 224, (cells // gx) * 224], dim=1).numpy()
    cls4 = [torch.nn.functional.normalize(centers.t() + 0.6 * torch.randn(D, 4, generator=g), dim=0) for _ in range(K)]
    cls2 = [c[:, :2].contiguous() for c in cls4]
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        scores4 = np.array([r_utils.rank_cls_score(torch.nn.functional.normalize(feats, dim=-1) @ c) for c in cls4])
        ens4 = r_utils.zero_shot_prompt_select(cls4, feats, 10, "cpu")
        ens2 = r_utils.zero_shot_prompt_select(cls2, feats, 10, "cpu")
        sub_label = int(r_sub.zero_shot_subtyping(ens4, feats, coords256, patch_size=256, overlap=True))
        sub_preds = r_sub.refine_seg(torch.softmax(torch.nn.functional.normalize(feats, dim=-1) @ ens4 * 10, 1),
                                     coords256, patch_size=256, overlap=True)
        det_ratio = float(r_det.zero_shot_detection(ens2, feats, coords256, patch_size=256, overlap=False))
        det_ratio_ov = float(r_det.zero_shot_detection(ens2, feats, coords256, patch_size=256, overlap=True))
        seg = r_seg.refine_seg(torch.softmax(torch.nn.functional.normalize(feats, dim=-1) @ ens2 * 10, 1),
                               coords224, patch_size=224, overlap=True)
    seg_keys = np.array([[int(s) for s in k.split("_")] for k in seg.keys()], dtype=np.int64)
    seg_vals = np.array(list(seg.values()), dtype=np.float64)
    sub_keys = np.array([[int(s) for s in k.split("_")] for k in sub_preds.keys()], dtype=np.int64)
    sub_vals = np.array(list(sub_preds.values()), dtype=np.int64)

    # oracle agreement
    assert abs(O.rank_cls_score(torch.nn.functional.normalize(feats, dim=-1) @ cls4[3]) - scores4[3]) < 1e-6
    assert torch.allclose(O.zero_shot_prompt_select(cls4, feats, 10), ens4, atol=1e-6)
    assert O.zero_shot_subtyping(ens4, feats, coords256, 256, True) == sub_label
    assert abs(O.zero_shot_detection(ens2, feats, coords256, 256, False) - det_ratio) < 1e-12
    assert abs(O.zero_shot_detection(ens2, feats, coords256, 256, True) - det_ratio_ov) < 1e-12
    keys, probs = O.zero_shot_segment_probs(ens2, feats, coords224, 224, True)
    assert np.array_equal(np.array(keys), seg_keys) and np.abs(probs - seg_vals).max() < 1e-6
    k2, mean = O.refine_mean_probs(torch.softmax(torch.nn.functional.normalize(feats, dim=-1) @ ens4 * 10, 1),
                                   coords256, 256, True)
    assert np.array_equal(np.array(k2), sub_keys) and np.array_equal(mean.argmax(1), sub_vals)
    print(f"[wsi] oracle == reference utils: label={sub_label} det={det_ratio:.4f}/{det_ratio_ov:.4f} "
          f"seg tiles={len(seg_vals)}")
    np.savez_compressed(os.path.join(GOLD, "wsi_logic.npz"),
                        feats=feats.numpy().astype(np.float32), coords256=coords256, coords224=coords224,
                        cls4=torch.stack(cls4).numpy(), topn=10,
                        scores4=scores4, ens4=ens4.numpy(), ens2=ens2.numpy(),
                        sub_label=sub_label, sub_keys=sub_keys, sub_preds=sub_vals,
                        det_ratio=det_ratio, det_ratio_overlap=det_ratio_ov,
                        seg_keys=seg_keys, seg_probs=seg_vals)


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    which = sys.argv[1:] or ["wsi", "bert2", "bert12", "vit2", "vit24"]
    if "wsi" in which:
        golden_wsi()
    if "bert2" in which:
        golden_bert(2, 4, seed=11)
    if "bert12" in which:
        golden_bert(12, 4, seed=12)
    if "vit2" in which:
        golden_vit(2, 3, seed=21)
    if "vit24" in which:
        golden_vit(24, 2, seed=22)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Pin oracle/keep_oracle.py against real implementations and write tests/golden/*.npz.

Runs ONLY in the build container (needs `transformers` and /root/reference):

  text tower   <- transformers.BertModel (the class the reference instantiates,
                  quick_start/keep_inference.py:49-50)
  image tower  <- transformers.Dinov2Model configured as ViT-L/16 + LayerScale
                  (independent implementation of timm vit_large_patch16_224's
                  block arithmetic; timm is not installed here), AND a module tree
                  shaped like timm's own (PatchEmbed / Attention with the fused qkv
                  reshape-permute / LayerScale / Mlp) on the ATen ops timm dispatches
                  (F.conv2d, F.layer_norm, F.scaled_dot_product_attention, F.gelu,
                  F.linear), loaded strictly from the release key layout
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


# --------------------------------------------------------------------------
# Second, independent pin of the image tower: the module tree timm's `vit_large_patch16_224` builds with the reference's ctor arguments
# (quick_start/keep_inference.py:32-40), written here from timm 1.0.15's published module semantics on the ATen ops timm dispatches:
# PatchEmbed = F.conv2d stride 16 -> flatten(2).transpose(1, 2); _pos_embed = cat(cls, x) + pos_embed; Block = x + ls1(attn(norm1(x))), x + ls2(mlp(norm2(x)));
# Attention = fused nn.Linear qkv -> reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4) -> F.scaled_dot_product_attention -> transpose(1, 2).reshape -> proj;
# LayerScale = x * gamma; Mlp = fc1 -> nn.GELU (erf) -> fc2; norm = nn.LayerNorm(eps 1e-6); global_pool 'token'; head Identity (num_classes = 0).
# Parameter names ARE the release's state_dict keys under `visual.`, so the seeded state_dict loads with strict=True -- which pins the
# key layout as well.  Not timm (absent offline), but fused-kernel ATen arithmetic instead of the oracle's hand-written matmul / softmax / erf formulas.
# --------------------------------------------------------------------------
class _TimmPatchEmbed(torch.nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.proj = torch.nn.Conv2d(3, dim, kernel_size=16, stride=16)

    def forward(self, x):
        return torch.nn.functional.conv2d(x, self.proj.weight, self.proj.bias, stride=16).flatten(2).transpose(1, 2)


class _TimmAttention(torch.nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads, self.head_dim = heads, dim // heads
        self.qkv = torch.nn.Linear(dim, dim * 3, bias=True)
        self.proj = torch.nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        x = torch.nn.functional.scaled_dot_product_attention(q, k, v)          # timm's fused_attn path, scale = head_dim ** -0.5
        return self.proj(x.transpose(1, 2).reshape(B, N, C))


class _TimmLayerScale(torch.nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.gamma = torch.nn.Parameter(torch.ones(dim))

    def forward(self, x):
        return x * self.gamma


class _TimmMlp(torch.nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1, self.act, self.fc2 = torch.nn.Linear(dim, hidden), torch.nn.GELU(), torch.nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class _TimmBlock(torch.nn.Module):
    def __init__(self, dim, heads, hidden):
        super().__init__()
        self.norm1, self.attn, self.ls1 = torch.nn.LayerNorm(dim, eps=1e-6), _TimmAttention(dim, heads), _TimmLayerScale(dim)
        self.norm2, self.mlp, self.ls2 = torch.nn.LayerNorm(dim, eps=1e-6), _TimmMlp(dim, hidden), _TimmLayerScale(dim)

    def forward(self, x):
        x = x + self.ls1(self.attn(self.norm1(x)))
        return x + self.ls2(self.mlp(self.norm2(x)))


class _TimmViT(torch.nn.Module):
    def __init__(self, depth, dim=1024, heads=16, hidden=4096, tokens=197):
        super().__init__()
        self.cls_token = torch.nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = torch.nn.Parameter(torch.zeros(1, tokens, dim))
        self.patch_embed = _TimmPatchEmbed(dim)
        self.blocks = torch.nn.Sequential(*[_TimmBlock(dim, heads, hidden) for _ in range(depth)])
        self.norm = torch.nn.LayerNorm(dim, eps=1e-6)

    def forward_tokens(self, x):
        x = self.patch_embed(x)
        x = torch.cat([self.cls_token.expand(x.shape[0], -1, -1), x], dim=1) + self.pos_embed
        return self.norm(self.blocks(x))

    def forward(self, x):
        return self.forward_tokens(x)[:, 0]                                       # global_pool = 'token', head = Identity


class _KeepImageSide(torch.nn.Module):
    """`visual` + `visual_head` of the reference's KEEPModel (keep_inference.py:32-46) and its encode_image (:54-58)."""

    def __init__(self, depth):
        super().__init__()
        self.visual = _TimmViT(depth)
        self.visual_head = torch.nn.Sequential(torch.nn.Linear(1024, 768), torch.nn.GELU(), torch.nn.Linear(768, 768))

    def encode_image(self, image_inputs):
        return torch.nn.functional.normalize(self.visual_head(self.visual(image_inputs)), dim=-1)


def aten_timm_from_sd(sd, depth):
    m = _KeepImageSide(depth).eval()
    m.load_state_dict({k: v for k, v in sd.items() if k.startswith("visual")}, strict=True)      # the release's key layout, strictly
    return m


def golden_vit(depth: int, batch: int, seed: int, name: str = None):
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
        at = aten_timm_from_sd(sd, depth)
        tok_at, feat_at = at.visual.forward_tokens(x), at.encode_image(x)
    d_tok = float((tok_or - tok_hf).abs().max())
    d_feat = float((feat_or - feat_hf).abs().max())
    a_tok, a_feat = float((tok_or - tok_at).abs().max()), float((feat_or - feat_at).abs().max())
    x_feat = float((feat_at - feat_hf).abs().max())
    print(f"[vit d{depth}] oracle vs Dinov2: max|dtok|={d_tok:.3e} max|dfeat|={d_feat:.3e}; oracle vs ATen-op timm restatement: max|dtok|={a_tok:.3e} "
          f"max|dfeat|={a_feat:.3e}; the two pins against each other: {x_feat:.3e}")
    assert d_tok < 5e-4 and d_feat < 2e-6, "oracle image tower disagrees with Dinov2-as-ViT-L"
    assert a_tok < 5e-4 and a_feat < 1e-6, "oracle image tower disagrees with the ATen-op restatement of timm's vit_large_patch16_224"
    np.savez_compressed(os.path.join(GOLD, name or f"vit_d{depth}.npz"),
                        depth=depth, batch=batch, weight_seed=seed, tile_seed=seed + 100,
                        tiles_checksum=checksum(x), qkv0_checksum=checksum(sd["visual.blocks.0.attn.qkv.weight"]),
                        cls=cls_hf.numpy(), features=feat_hf.numpy(), features_aten_timm=feat_at.numpy(),
                        oracle_dtok=d_tok, oracle_dfeat=d_feat, oracle_dtok_aten_timm=a_tok, oracle_dfeat_aten_timm=a_feat, pins_dfeat=x_feat)


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


def golden_wsi_callers(seed: int = 9):
    """Rows a9 / a10 / a17 pinned to the reference's OWN functions: `zero_shot_classifier`, `get_zeroshot_classifier`
    (WSI_evaluation/utils.py:64-104) through a stand-in KEEP_model whose text tower is transformers.BertModel -- the class the
    reference instantiates -- on seeded weights and a deterministic stand-in tokenizer; and the three `run(classifier,
    dataloader, device)` loops (subtyping_utils.py:12, detection_utils.py:12, segment_utils.py:16) over the reference's
    `WSI_Classification_Dataset` (utils.py:11-61) on .pt feature files and a torch DataLoader."""
    import contextlib, io, tempfile
    import pandas as pd
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from standins import (CALLER_DIAG_MAP, CALLER_LABEL_MAP, CALLER_PROMPTS, CALLER_SLIDES, HashTokenizer, caller_slide_features)
    r_utils, r_sub, r_det, r_seg = import_reference_wsi()
    sd = synth_state_dict(small_shape(1, 2), seed=seed, vision=False)
    bert = hf_bert_from_sd(sd, 2, "eager")

    class RefTextModel:                           # keep_inference.py:60-62 on the HF module
        def encode_text(self, text_inputs):
            return torch.nn.functional.normalize(bert(**text_inputs).pooler_output, dim=-1)

    KEEP_model = {"model": RefTextModel(), "tokenizer": HashTokenizer()}
    out = {}
    with torch.no_grad():
        for i, p in enumerate(CALLER_PROMPTS):
            out[f"cls_normal_{i}"] = r_utils.get_zeroshot_classifier(KEEP_model, CALLER_LABEL_MAP, p, "cpu", add_normal=True).numpy()
            out[f"cls_plain_{i}"] = r_utils.get_zeroshot_classifier(KEEP_model, CALLER_LABEL_MAP, p, "cpu").numpy()
        names = ["lung adenocarcinoma", "normal tissue"]
        out["zsc_str"] = r_utils.zero_shot_classifier(KEEP_model, names, "an H&E image of CLASSNAME.", "cpu").numpy()
        out["zsc_list"] = r_utils.zero_shot_classifier(KEEP_model, names, ["CLASSNAME.", "a photo of CLASSNAME."], "cpu").numpy()
        # oracle agreement: column c = renormalised embedding of the filled (first) template
        ref = O.encode_text(sd, HashTokenizer()(["an H&E image of normal tissue."]))[0]
        assert (torch.from_numpy(out["zsc_str"][:, 1]) - ref / ref.norm()).abs().max() < 2e-6
    assert out["cls_normal_0"].shape == (768, 4) and out["cls_plain_0"].shape == (768, 3)

    feats = caller_slide_features()
    g = torch.Generator().manual_seed(seed + 1)
    cls3 = torch.nn.functional.normalize(torch.randn(768, 3, generator=g), dim=0)
    cls2 = cls3[:, :2].contiguous()
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "pt_files"))
        for sid, f in feats.items():
            torch.save(f, os.path.join(tmp, "pt_files", sid + ".pt"))
        df = pd.DataFrame([{"slide_id": sid, "Diagnosis": d} for sid, _, d in CALLER_SLIDES])
        ds = r_utils.WSI_Classification_Dataset(df, tmp, use_h5=False, label_map=CALLER_DIAG_MAP)
        dl = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False)
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            sub_logits, sub_coords, sub_targets = r_sub.run(cls3, dl, "cpu")
            det_probs, _, det_targets = r_det.run(cls2, dl, "cpu")
            seg_probs, seg_coords = r_seg.run(cls2, dl, "cpu")
    assert sub_targets == det_targets == {sid: CALLER_DIAG_MAP[d] for sid, _, d in CALLER_SLIDES}
    assert all(c == [] for c in sub_coords.values()) and list(seg_coords) == [sid for sid, _, _ in CALLER_SLIDES]
    for sid, f in feats.items():
        refl = O.l2_normalize(f) @ cls3
        assert (sub_logits[sid] - refl).abs().max() < 1e-6
        assert (det_probs[sid] - O.sim_softmax(O.l2_normalize(f) @ cls2, 10.0)).abs().max() < 1e-6
        out[f"run_sub_{sid}"] = sub_logits[sid].numpy()
        out[f"run_det_{sid}"] = det_probs[sid].numpy()
        out[f"run_seg_{sid}"] = seg_probs[sid].numpy()
    print(f"[wsi_callers] reference zero_shot_classifier / get_zeroshot_classifier / run x3 recorded ({len(out)} arrays)")
    np.savez_compressed(os.path.join(GOLD, "wsi_callers.npz"), weight_seed=seed, feat_seed=77, cls3=cls3.numpy(),
                        feats_checksum=float(sum(checksum(f) for f in feats.values())), **out)


def golden_signatures():
    """The L4 call surface as the reference declares it (inspect.signature of the live functions) -> tests/golden/reference_signatures.json;
    tests/test_reference_signatures.py holds keep_amd's functions to it."""
    import inspect, json
    mods = dict(zip(("utils", "subtyping_utils", "detection_utils", "segment_utils"), import_reference_wsi()))
    want = {"utils": ["zero_shot_classifier", "get_zeroshot_classifier", "rank_cls_score", "zero_shot_prompt_select", "cood2str", "str2cood", "accuracy"],
            "subtyping_utils": ["run", "refine_seg", "zero_shot_subtyping"],
            "detection_utils": ["run", "refine_seg", "zero_shot_detection"],
            "segment_utils": ["run", "zero_shot_segment", "refine_seg"]}
    table = {}
    for mod, names in want.items():
        for n in names:
            fn = getattr(mods[mod], n)
            params = [[p.name] if p.default is inspect.Parameter.empty else [p.name, p.default]
                      for p in inspect.signature(fn).parameters.values()]
            table[f"{mod}.{n}"] = {"params": params, "line": f"WSI_evaluation/{mod}.py:{inspect.getsourcelines(fn)[1]}"}
    with open(os.path.join(GOLD, "reference_signatures.json"), "w") as f:
        json.dump(table, f, indent=1)
    print(f"[signatures] {len(table)} reference functions recorded")


def import_reference_tile_eval():
    """training/path_training/zero_shot.py as shipped.  Its package imports (`path_open_clip`: timm/open_clip model
    code that does not import here) are replaced by a stub package that carries the REAL metric functions of
    training/path_open_clip/zeroshot_metrics.py; `.precision` is the real module."""
    import importlib.util

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod

    tr = os.path.join(REF, "training")
    metrics = load("_ref_zeroshot_metrics", os.path.join(tr, "path_open_clip", "zeroshot_metrics.py"))
    poc = types.ModuleType("path_open_clip")
    poc.__path__ = []
    for n in ("get_input_dtype", "get_tokenizer", "build_zero_shot_classifier"):
        setattr(poc, n, lambda *a, **k: None)
    poc.IMAGENET_CLASSNAMES, poc.OPENAI_IMAGENET_TEMPLATES = [], []
    poc.retrieval_metrics, poc.classification_metrics = metrics.retrieval_metrics, metrics.classification_metrics
    tok = types.ModuleType("path_open_clip.tokenizer")
    tok.tokenize = lambda *a, **k: None
    sys.modules["path_open_clip"], sys.modules["path_open_clip.tokenizer"] = poc, tok
    pkg = types.ModuleType("path_training")
    pkg.__path__ = [os.path.join(tr, "path_training")]
    sys.modules["path_training"] = pkg
    load("path_training.precision", os.path.join(tr, "path_training", "precision.py"))
    return load("path_training.zero_shot", os.path.join(tr, "path_training", "zero_shot.py"))


def golden_tile_eval(seed: int = 9):
    """Runs the reference's zero_shot_eval on synthetic embeddings through a stand-in model object (encode_image /
    encode_text / __call__ return pre-drawn features; the encoders themselves are pinned elsewhere) and records its
    result dict."""
    import json, tempfile, warnings
    from types import SimpleNamespace as NS
    zs = import_reference_tile_eval()
    g = torch.Generator().manual_seed(seed)
    D, C, N, R = 768, 4, 403, 260
    names = ["Benign", "InSitu", "Invasive", "Normal"]
    centers = torch.nn.functional.normalize(torch.randn(C, D, generator=g), dim=-1)
    lab = torch.randint(0, C, (N,), generator=g)
    img = centers[lab] * 1.3 + torch.randn(N, D, generator=g) * 0.1               # un-normalised image features
    caps = {n: centers[c][None] * 0.6 + torch.randn(50, D, generator=g) * 0.11 for c, n in enumerate(names)}
    prompts = {str(i): {"classnames": {n: f"{n.lower()} tissue v{i % 7}" for n in names}, "templates": f"a photo {i} of CLASSNAME."}
               for i in range(50)}
    ret_img = torch.randn(R, D, generator=g)
    ret_txt = ret_img * 0.055 + torch.randn(R, D, generator=g)                     # p@10 / p@50 strictly inside (0, 1)
    labels = [names[int(c)] for c in lab]

    class Tok:                                   # BatchEncoding stand-in (not a dict: zero_shot.py:116 branches on that)
        def __init__(self, texts):
            self.texts = texts

        def __getitem__(self, k):
            return getattr(self, k)

        def to(self, **kw):
            return self

    cap_lookup = {}
    for n in names:
        for i in range(50):
            cap_lookup[prompts[str(i)]["templates"].replace("CLASSNAME", prompts[str(i)]["classnames"][n])] = caps[n][i]
    txt_lookup = {f"caption {i}": ret_txt[i] for i in range(R)}

    def bert_tok(texts, **kw):
        return Tok(texts=list(texts))

    class Model:
        def eval(self):
            return self

        def encode_image(self, images):
            return images                         # "images" are the pre-drawn features

        def encode_text(self, inp):
            return torch.stack([cap_lookup[t] for t in inp["texts"]])

        def __call__(self, images, inp):
            return {"image_features": images, "text_features": torch.stack([txt_lookup[t] for t in inp["texts"]])}

    def loader(pairs, bs):
        return NS(dataloader=[(torch.stack([p[0] for p in pairs[i:i + bs]]), [p[1] for p in pairs[i:i + bs]])
                              for i in range(0, len(pairs), bs)])

    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        json.dump(prompts, f)
    cfg = NS(SOLVER=NS(ZEROSHOT_FREQUENCY=1, EPOCHS=1), MODEL=NS(PRECISION="fp32", KNOWLEDGE_GUIDANCE=False, BERT_PRETRAIN="x",
                                                                 TEXT_ENCODER="bert"),
             DATASET=NS(ZEROSHOT_CLS_PROMPTS=f.name), DATALOADER=NS(BATCH_SIZE=64))
    args = NS(distributed=False, horovod=False, device="cpu")
    data = {"zeroshot_cls": loader(list(zip(img, labels)), 64),
            "zeroshot_ret": loader([(ret_img[i], f"caption {i}") for i in range(R)], 32)}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = zs.zero_shot_eval(Model(), {"bert": bert_tok}, data, 1, args, cfg)
        ref_caps = zs.label2cap(cfg)
    os.unlink(f.name)
    # oracle agreement
    assert O.label2cap(prompts) == ref_caps
    val = O.tile_classification_rounds(img.numpy(), {n: caps[n].numpy() for n in names}, labels)
    q = O.wf1_quartiles(val)
    ret = O.retrieval_p_at_k(ret_img.numpy(), ret_txt.numpy())
    for k in q:
        assert abs(q[k] - res[k]) < 1e-12, (k, q[k], res[k])
    assert ret["p@10"] == res["zeroshot-ret-p@10"] and ret["p@50"] == res["zeroshot-ret-p@50"]
    print(f"[tile_eval] oracle == reference zero_shot_eval: {res}")
    np.savez_compressed(os.path.join(GOLD, "tile_eval.npz"),
                        img=img.numpy(), labels=np.array(labels), names=np.array(names),
                        caps=np.stack([caps[n].numpy() for n in names]), prompts=json.dumps(prompts),
                        ret_img=ret_img.numpy(), ret_txt=ret_txt.numpy(), wf1_rounds=val,
                        wf1_median=res["zeroshot-cls-WF1-median"], wf1_q1=res["zeroshot-cls-WF1-Q1"], wf1_q3=res["zeroshot-cls-WF1-Q3"],
                        p10=res["zeroshot-ret-p@10"], p50=res["zeroshot-ret-p@50"])


def golden_c3(n_tiles: int = 4096, chunk: int = 256, n_prompts: int = 64):
    """BASELINE.json config 3 (full dual tower: 4096 tiles x 64 prompts, sim matrix + argmax) on the bench weights
    (seed 0).  Oracle only (it is pinned against BertModel / Dinov2Model by the sections above).  Tiles are
    regenerated from per-chunk seeds on the GPU box, so only the expected outputs are stored: the [4096,64] fp32 cosine
    matrix, its row argmax, the text features, and the image features of the first chunk."""
    import time
    sd = synth_state_dict(KEEPShape(), seed=0)
    toks = synth_prompts(n_prompts, 256, seed=1)
    feats = []
    t0 = time.time()
    with torch.no_grad():
        txt = O.encode_text(sd, toks)
        for c in range(n_tiles // chunk):
            ck = f"/tmp/c3_chunk_{c}.pt"              # resumable: ~90 s of CPU per chunk
            if os.path.exists(ck):
                feats.append(torch.load(ck))
                continue
            x = synth_tiles(chunk, seed=5000 + c)
            part = torch.cat([O.encode_image(sd, x[i:i + 32]) for i in range(0, chunk, 32)])
            torch.save(part, ck)
            feats.append(part)
            print(f"[c3] chunk {c + 1}/{n_tiles // chunk}  {time.time() - t0:.0f}s", flush=True)
    img = torch.cat(feats)
    sims = img @ txt.t()
    top2 = sims.topk(2, dim=1).values
    np.savez_compressed(os.path.join(GOLD, "c3_dual_tower.npz"),
                        n_tiles=n_tiles, chunk=chunk, tile_seed0=5000, weight_seed=0, prompt_seed=1,
                        tiles0_checksum=checksum(synth_tiles(chunk, seed=5000)),
                        input_ids=toks["input_ids"].numpy().astype(np.int32),
                        attention_mask=toks["attention_mask"].numpy().astype(np.int8),
                        txt=txt.numpy(), img_first=img[:chunk].numpy(),
                        sims=sims.numpy(), argmax=sims.argmax(1).numpy().astype(np.int16),
                        margin=(top2[:, 0] - top2[:, 1]).numpy())
    print(f"[c3] {n_tiles} x {n_prompts}: min top-2 margin {float((top2[:, 0] - top2[:, 1]).min()):.3e}")


def golden_families(n_tiles: int = 96, n_prompts: int = 64, seed: int = 3):
    """The two extra weight families of keep_amd.synth (heavy-tailed activations / outlier LayerNorm gains; tiny LayerScale) at full
    depth through both oracle towers -> tests/golden/family_<name>.npz: the [n_tiles, n_prompts] cosine matrix, argmax, top-2 margins.
    The oracle is pinned by the vit / bert sections; weights and tiles are regenerated from seeds on the GPU box."""
    toks = synth_prompts(n_prompts, 256, seed=seed + 40)
    for family in ("heavy_tail", "small_ls"):
        sd = synth_state_dict(KEEPShape(), seed=seed, family=family)
        x = synth_tiles(n_tiles, seed=700 + seed)
        with torch.no_grad():
            txt = O.encode_text(sd, toks)
            img = torch.cat([O.encode_image(sd, x[i:i + 32]) for i in range(0, n_tiles, 32)])
            tok = O.vit_tokens(sd, x[:4], 24)
        assert bool(torch.isfinite(img).all())
        sims = img @ txt.t()
        top2 = sims.topk(2, dim=1).values
        print(f"[family {family}] |token| max {float(tok.abs().max()):.1f} median {float(tok.abs().median()):.3f}; tile-tile cosine "
              f"{float((img @ img.t())[0, 1]):.4f}; min top-2 margin {float((top2[:, 0] - top2[:, 1]).min()):.2e}")
        np.savez_compressed(os.path.join(GOLD, f"family_{family}.npz"), family=family, weight_seed=seed, tile_seed=700 + seed, n_tiles=n_tiles,
                            tiles_checksum=checksum(x), qkv0_checksum=checksum(sd["visual.blocks.0.attn.qkv.weight"]),
                            input_ids=toks["input_ids"].numpy().astype(np.int32), attention_mask=toks["attention_mask"].numpy().astype(np.int8),
                            sims=sims.numpy(), argmax=sims.argmax(1).numpy().astype(np.int16), margin=(top2[:, 0] - top2[:, 1]).numpy())


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    which = sys.argv[1:] or ["wsi", "wsi_callers", "signatures", "tile_eval", "bert2", "bert12", "vit2", "vit24", "vit24_bench"]
    if "tile_eval" in which:
        golden_tile_eval()
    if "wsi" in which:
        golden_wsi()
    if "wsi_callers" in which:
        golden_wsi_callers()
    if "signatures" in which:
        golden_signatures()
    if "families" in which:
        golden_families()
    if "bert2" in which:
        golden_bert(2, 4, seed=11)
    if "bert12" in which:
        golden_bert(12, 4, seed=12)
    if "vit2" in which:
        golden_vit(2, 3, seed=21)
    if "vit24" in which:
        golden_vit(24, 2, seed=22)
    if "vit24_bench" in which:          # the weights bench.py runs on (seed 0): lets the bench print a live parity figure
        golden_vit(24, 8, seed=0, name="vit_d24_bench.npz")
    if "c3" in which:                   # ~15 min of CPU: not in the default list
        golden_c3()


if __name__ == "__main__":
    main()

"""CPU oracle for the KEEP zero-shot hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain fp32 PyTorch-CPU restatement of the arithmetic behind
``KEEPModel.encode_image`` / ``encode_text`` (reference
``quick_start/keep_inference.py:54-62``), the similarity at ``:104`` and the
WSI zero-shot host logic in ``WSI_evaluation/{utils,subtyping_utils,
detection_utils,segment_utils}.py``.  Only ``tests/``, ``__graft_entry__.smoke``
and ``bench.py``'s ``cpu_baseline`` leg may import it; the product package
``keep_amd`` never does (tests/test_layout.py enforces that).

Pinning.  The two towers live in third-party packages that are not vendored in
the reference: ``timm==1.0.15`` (absent here) and ``transformers==4.34.0``
(``training/requirements.txt:13,17``).  ``tools/make_golden.py`` pins this
restatement, in the build container, against
  * ``transformers.BertModel`` (the reference's own text tower class,
    keep_inference.py:49-50) on seeded synthetic weights,
  * ``transformers.Dinov2Model`` configured as ViT-L/16 + LayerScale -- an
    independent implementation of the same block arithmetic (timm itself is
    not installed, so the image tower is pinned against this stand-in, not
    against timm: "image-tower parity pinned to an independent implementation"),
  * the reference's ``WSI_evaluation/*utils.py`` imported from /root/reference,
and commits the resulting vectors under ``tests/golden/``.  The reference ships
no tests or golden vectors of its own.
"""
from __future__ import annotations

import math
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------
# operand rounding model (used only to predict the HIP path's error budget)
# --------------------------------------------------------------------------
def _linear(x: Tensor, w: Tensor, b: Optional[Tensor], operand_dtype=None) -> Tensor:
    """y = x W^T + b in fp32; optionally round both GEMM operands to
    ``operand_dtype`` first (fp32 accumulate), which is the MFMA error model."""
    if operand_dtype is not None:
        x = x.to(operand_dtype).to(torch.float32)
        w = w.to(operand_dtype).to(torch.float32)
    y = x @ w.t()
    return y if b is None else y + b


def gelu_erf(x: Tensor) -> Tensor:
    """Exact GELU: timm ``nn.GELU`` and HF ``hidden_act='gelu'`` are both erf-based."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc * torch.rsqrt(var + eps) * w + b


def l2_normalize(x: Tensor, eps: float = 1e-12) -> Tensor:
    """``F.normalize(x, dim=-1)``: x / max(||x||, eps)  (keep_inference.py:56,61)."""
    return x / x.norm(dim=-1, keepdim=True).clamp_min(eps)


# --------------------------------------------------------------------------
# image tower: timm vit_large_patch16_224 as built at keep_inference.py:32-40
# (semantics: SURVEY.md §A.1)
# --------------------------------------------------------------------------
def patchify(x: Tensor, patch: int = 16) -> Tensor:
    """[B,3,H,W] -> [B, (H/p)*(W/p), 3*p*p], inner order (c, ph, pw): the im2col
    that turns the stride-16 conv of ``PatchEmbed`` into a GEMM."""
    B, C, H, W = x.shape
    gh, gw = H // patch, W // patch
    x = x.reshape(B, C, gh, patch, gw, patch).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(B, gh * gw, C * patch * patch)


def _sdpa(q: Tensor, k: Tensor, v: Tensor, bias: Optional[Tensor]) -> Tensor:
    """softmax(q k^T / sqrt(hd) + bias) v, per head; q,k,v [B,H,T,hd]."""
    s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(q.shape[-1]))
    if bias is not None:
        s = s + bias
    return torch.softmax(s, dim=-1) @ v


def vit_tokens(sd: Mapping[str, Tensor], x: Tensor, depth: int, heads: int = 16,
               eps: float = 1e-6, operand_dtype=None, prefix: str = "visual.") -> Tensor:
    """Token stream after the final LayerNorm, [B,197,D]."""
    x = x.to(torch.float32)
    B = x.shape[0]
    wpe = sd[prefix + "patch_embed.proj.weight"]
    D = wpe.shape[0]
    patch = wpe.shape[-1]
    p = _linear(patchify(x, patch), wpe.reshape(D, -1), sd[prefix + "patch_embed.proj.bias"], operand_dtype)
    t = torch.cat([sd[prefix + "cls_token"].expand(B, -1, -1), p], dim=1) + sd[prefix + "pos_embed"]
    N = t.shape[1]
    hd = D // heads
    for i in range(depth):
        bp = f"{prefix}blocks.{i}."
        h = layer_norm(t, sd[bp + "norm1.weight"], sd[bp + "norm1.bias"], eps)
        qkv = _linear(h, sd[bp + "attn.qkv.weight"], sd[bp + "attn.qkv.bias"], operand_dtype)
        qkv = qkv.reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        if operand_dtype is not None:   # attention GEMM operands are rounded too
            q, k, v = (z.to(operand_dtype).to(torch.float32) for z in (q, k, v))
        a = _sdpa(q, k, v, None).transpose(1, 2).reshape(B, N, D)
        t = t + sd[bp + "ls1.gamma"] * _linear(a, sd[bp + "attn.proj.weight"], sd[bp + "attn.proj.bias"], operand_dtype)
        h = layer_norm(t, sd[bp + "norm2.weight"], sd[bp + "norm2.bias"], eps)
        m = gelu_erf(_linear(h, sd[bp + "mlp.fc1.weight"], sd[bp + "mlp.fc1.bias"], operand_dtype))
        t = t + sd[bp + "ls2.gamma"] * _linear(m, sd[bp + "mlp.fc2.weight"], sd[bp + "mlp.fc2.bias"], operand_dtype)
    return layer_norm(t, sd[prefix + "norm.weight"], sd[prefix + "norm.bias"], eps)


def vit_forward(sd, x, depth, heads=16, eps=1e-6, operand_dtype=None) -> Tensor:
    """``self.visual(x)``: CLS pooling (global_pool='token', num_classes=0) -> [B,D]."""
    return vit_tokens(sd, x, depth, heads, eps, operand_dtype)[:, 0]


def visual_head(sd, f: Tensor) -> Tensor:
    """``nn.Sequential(Linear, GELU, Linear)`` -- keep_inference.py:42-46."""
    h = gelu_erf(_linear(f, sd["visual_head.0.weight"], sd["visual_head.0.bias"]))
    return _linear(h, sd["visual_head.2.weight"], sd["visual_head.2.bias"])


def count_vit_depth(sd) -> int:
    n = 0
    while f"visual.blocks.{n}.norm1.weight" in sd:
        n += 1
    return n


def encode_image(sd, x: Tensor, heads: int = 16, operand_dtype=None) -> Tensor:
    """keep_inference.py:54-58."""
    f = vit_forward(sd, x, count_vit_depth(sd), heads, 1e-6, operand_dtype)
    return l2_normalize(visual_head(sd, f))


# --------------------------------------------------------------------------
# text tower: HF BertModel(BertConfig(**text_config)) -- keep_inference.py:49-50
# (semantics: SURVEY.md §A.2)
# --------------------------------------------------------------------------
def count_bert_layers(sd) -> int:
    n = 0
    while f"text.encoder.layer.{n}.attention.self.query.weight" in sd:
        n += 1
    return n


def bert_pooled(sd, input_ids: Tensor, token_type_ids: Optional[Tensor], attention_mask: Optional[Tensor],
                heads: int = 12, eps: float = 1e-12, operand_dtype=None, prefix: str = "text.") -> Tensor:
    """``self.text(**inputs).pooler_output`` -> [P,H]."""
    P, T = input_ids.shape
    if token_type_ids is None:
        token_type_ids = torch.zeros_like(input_ids)
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids)
    e = (sd[prefix + "embeddings.word_embeddings.weight"][input_ids]
         + sd[prefix + "embeddings.token_type_embeddings.weight"][token_type_ids]
         + sd[prefix + "embeddings.position_embeddings.weight"][:T][None])
    h = layer_norm(e, sd[prefix + "embeddings.LayerNorm.weight"], sd[prefix + "embeddings.LayerNorm.bias"], eps)
    H = h.shape[-1]
    hd = H // heads
    # key-padding mask only; HF adds finfo.min to masked keys
    bias = (1.0 - attention_mask[:, None, None, :].to(torch.float32)) * torch.finfo(torch.float32).min
    L = count_bert_layers(sd)
    for i in range(L):
        lp = f"{prefix}encoder.layer.{i}."

        def heads_of(z):
            z = z.reshape(P, T, heads, hd).transpose(1, 2)
            return z.to(operand_dtype).to(torch.float32) if operand_dtype is not None else z

        q = heads_of(_linear(h, sd[lp + "attention.self.query.weight"], sd[lp + "attention.self.query.bias"], operand_dtype))
        k = heads_of(_linear(h, sd[lp + "attention.self.key.weight"], sd[lp + "attention.self.key.bias"], operand_dtype))
        v = heads_of(_linear(h, sd[lp + "attention.self.value.weight"], sd[lp + "attention.self.value.bias"], operand_dtype))
        a = _sdpa(q, k, v, bias).transpose(1, 2).reshape(P, T, H)
        o = _linear(a, sd[lp + "attention.output.dense.weight"], sd[lp + "attention.output.dense.bias"], operand_dtype)
        h = layer_norm(h + o, sd[lp + "attention.output.LayerNorm.weight"], sd[lp + "attention.output.LayerNorm.bias"], eps)
        m = gelu_erf(_linear(h, sd[lp + "intermediate.dense.weight"], sd[lp + "intermediate.dense.bias"], operand_dtype))
        o = _linear(m, sd[lp + "output.dense.weight"], sd[lp + "output.dense.bias"], operand_dtype)
        h = layer_norm(h + o, sd[lp + "output.LayerNorm.weight"], sd[lp + "output.LayerNorm.bias"], eps)
    return torch.tanh(_linear(h[:, 0], sd[prefix + "pooler.dense.weight"], sd[prefix + "pooler.dense.bias"]))


def encode_text(sd, text_inputs: Mapping[str, Tensor], heads: int = 12, operand_dtype=None) -> Tensor:
    """keep_inference.py:60-62."""
    pooled = bert_pooled(sd, text_inputs["input_ids"], text_inputs.get("token_type_ids"),
                         text_inputs.get("attention_mask"), heads, 1e-12, operand_dtype)
    return l2_normalize(pooled)


# --------------------------------------------------------------------------
# similarity and its per-tile reductions
# --------------------------------------------------------------------------
def similarity(img: Tensor, txt: Tensor, scale: float = 1.0) -> Tensor:
    """``img_feature @ text_feature.T`` (keep_inference.py:104), optionally scaled."""
    return scale * (img.to(torch.float32) @ txt.to(torch.float32).t())


def sim_argmax(sim: Tensor) -> Tensor:
    """Row argmax, lowest index wins ties (torch CPU semantics)."""
    return sim.argmax(dim=1).to(torch.int32)


def sim_softmax(sim: Tensor, scale: float = 10.0) -> Tensor:
    """``softmax(logits*10, 1)`` -- subtyping_utils.py:72, detection_utils.py:93, segment_utils.py:49."""
    return torch.softmax(sim * scale, dim=1)


# --------------------------------------------------------------------------
# WSI zero-shot host logic (WSI_evaluation/*.py)
# --------------------------------------------------------------------------
def rank_cls_score(logits: Tensor) -> float:
    """utils.py:107-117: mean over tiles of (v1 - v2) - |v1 + v2 - 1| on raw cosine logits."""
    v = torch.topk(logits, k=2, dim=1).values
    return float(((v[:, 0] - v[:, 1]) - (v[:, 0] + v[:, 1] - 1).abs()).mean())


def zero_shot_prompt_select(classifiers: Sequence[Tensor], tile_features: Tensor, topn: int) -> Tensor:
    """utils.py:119-146."""
    f = l2_normalize(tile_features.to(torch.float32))
    scores = [rank_cls_score(f @ c) for c in classifiers]
    # torch.sort(descending=True) on CPU is not guaranteed stable; the reference
    # inherits whatever order torch picks for exact ties.
    order = torch.sort(torch.tensor(scores), descending=True).indices
    merged = torch.zeros_like(classifiers[0])
    for i in order[:topn]:
        merged = merged + classifiers[int(i)]
    return F.normalize(merged, p=2, dim=0)


def random_prompt_ensemble(classifiers: Sequence[Tensor], topn: int) -> Tensor:
    """zeroshot_subtyping_WSI.py:68-76 (same in the detection / segmentation scripts): unscreened ensemble."""
    import random
    acc = torch.zeros_like(classifiers[-1])
    cter = 0
    while cter < topn:
        random.seed(cter)
        acc = acc + classifiers[random.randint(0, len(classifiers) - 1)]
        cter += 1
    return F.normalize(acc, p=2, dim=0)


def _dedupe_first(coords: np.ndarray) -> Tuple[Dict[Tuple[int, int], int], List[Tuple[int, int]]]:
    first: Dict[Tuple[int, int], int] = {}
    for i, c in enumerate(coords):
        key = (int(c[0]), int(c[1]))
        if key not in first:
            first[key] = i
    return first, list(first.keys())


def refine_mean_probs(probs: Tensor, coords, patch_size: int, overlap: bool) -> Tuple[List[Tuple[int, int]], np.ndarray]:
    """Common core of the three ``refine_seg`` variants (subtyping_utils.py:38-65,
    detection_utils.py:39-74, segment_utils.py:63-89): de-duplicate coordinates
    (first occurrence wins), then for each tile average the probability rows of
    the existing tiles among {(x-p,y-p),(x,y-p),(x-p,y),(x,y)}.  Returns the
    unique coords in first-seen order and the float32 mean rows (numpy mean of
    float32 rows stays float32, as in the reference)."""
    coords = np.asarray(coords)
    p = probs.detach().cpu().numpy().astype(np.float32)
    first, keys = _dedupe_first(coords)
    out = np.empty((len(keys), p.shape[1]), dtype=np.float32)
    for j, (x, y) in enumerate(keys):
        if overlap:
            rows = [p[first[k]] for k in ((x - patch_size, y - patch_size), (x, y - patch_size),
                                          (x - patch_size, y), (x, y)) if k in first]
            out[j] = np.array(rows).mean(0)
        else:
            out[j] = p[first[(x, y)]]
    return keys, out


def zero_shot_subtyping(classifier: Tensor, tile_features: Tensor, tile_coords, patch_size: int = 256,
                        overlap: bool = True) -> int:
    """subtyping_utils.py:67-83: slide label = most frequent refined tile label among
    the non-'Normal' (all but last) classes."""
    probs = sim_softmax(l2_normalize(tile_features.to(torch.float32)) @ classifier, 10.0)
    keys, mean = refine_mean_probs(probs, tile_coords, patch_size, overlap)
    pred = torch.from_numpy(mean).max(1).indices.numpy()
    C = classifier.shape[1]
    frac = [(pred == c).sum() / len(pred) for c in range(C)]
    return int(torch.tensor(frac[0:-1]).max(0).indices)


def zero_shot_detection(classifier: Tensor, tile_features: Tensor, tile_coords, patch_size: int = 256,
                        overlap: bool = False, threshold: float = 0.5) -> float:
    """detection_utils.py:88-100: fraction of (unique) tiles whose refined tumour
    probability exceeds 0.5."""
    probs = sim_softmax(l2_normalize(tile_features.to(torch.float32)) @ classifier, 10.0)
    keys, mean = refine_mean_probs(probs, tile_coords, patch_size, overlap)
    return float((mean[:, 1] > threshold).sum() / len(keys))


def zero_shot_segment_probs(classifier: Tensor, tile_features: Tensor, tile_coords, patch_size: int = 224,
                            overlap: bool = True) -> Tuple[List[Tuple[int, int]], np.ndarray]:
    """segment_utils.py:44-52 + :63-89: dense per-tile tumour probability map."""
    probs = sim_softmax(l2_normalize(tile_features.to(torch.float32)) @ classifier, 10.0)
    keys, mean = refine_mean_probs(probs, tile_coords, patch_size, overlap)
    return keys, mean[:, 1]


def build_classifier(text_embeddings: Tensor) -> Tensor:
    """utils.py:76-83 for single-string templates: per class normalise -> mean over the
    (one) template -> renormalise, stack on dim=1 -> [768, C]."""
    cols = []
    for e in text_embeddings:
        e = F.normalize(e[None], dim=-1).mean(0)
        cols.append(e / e.norm())
    return torch.stack(cols, dim=1)


# --------------------------------------------------------------------------
# tile-level zero-shot evaluation protocol (training/path_training/zero_shot.py:91-139, 141-176, 215-232;
# metrics training/path_open_clip/zeroshot_metrics.py:6-17, :31).  numpy, as the reference.
# --------------------------------------------------------------------------
def label2cap(prompts: Mapping) -> Dict[str, List[str]]:
    """zero_shot.py:49-62 on the parsed prompt JSON: per class name, the 50 filled captions."""
    out: Dict[str, List[str]] = {}
    for type_name in list(prompts["0"]["classnames"].keys()):
        out[type_name] = [prompts[str(i)]["templates"].replace("CLASSNAME", prompts[str(i)]["classnames"][type_name])
                          for i in range(50)]
    return out


def weighted_f1(y_true: Sequence, y_pred: Sequence) -> float:
    """sklearn.metrics.f1_score(average='weighted') (zeroshot_metrics.py:31): per-class F1 over the labels present
    in y_true or y_pred, 0 where undefined, weighted by the class's count in y_true."""
    labels = sorted(set(y_true) | set(y_pred))
    yt, yp = np.asarray(y_true), np.asarray(y_pred)
    f, w = [], []
    for c in labels:
        tp = float(np.sum((yt == c) & (yp == c)))
        fp = float(np.sum((yt != c) & (yp == c)))
        fn = float(np.sum((yt == c) & (yp != c)))
        den = 2 * tp + fp + fn
        f.append(2 * tp / den if den else 0.0)
        w.append(tp + fn)
    return float(np.average(np.array(f), weights=np.array(w)))


def tile_classification_rounds(image_embeddings: np.ndarray, cap_embeddings: Mapping[str, np.ndarray],
                               label_list: Sequence) -> np.ndarray:
    """zero_shot.py:118-139: image features re-normalised in numpy; for each of the 50 prompt rounds the class
    embeddings are row i of every class's caption embeddings, normalised; prediction = argmax of the float32
    dot products; WF1 per round."""
    img = np.array(image_embeddings)
    img = img / np.linalg.norm(img, axis=1, keepdims=True)
    names = list(cap_embeddings.keys())
    out = []
    for i in range(50):
        rnd = np.array([cap_embeddings[n][i, :] for n in names])
        rnd = rnd / np.linalg.norm(rnd, axis=1, keepdims=True)
        score = img.dot(rnd.T)
        pred = [names[int(np.argmax(s))] for s in score]
        out.append(weighted_f1(list(label_list), pred))
    return np.array(out)


def wf1_quartiles(val_cls: np.ndarray) -> Dict[str, float]:
    """zero_shot.py:216-222 (np.percentile(..., interpolation='midpoint'))."""
    q1, med, q3 = np.percentile(val_cls, (25, 50, 75), method="midpoint")
    return {"zeroshot-cls-WF1-median": float(med), "zeroshot-cls-WF1-Q1": float(q1), "zeroshot-cls-WF1-Q3": float(q3)}


def retrieval_p_at_k(image_embeddings: np.ndarray, text_embeddings: np.ndarray) -> Dict[str, float]:
    """zero_shot.py:161-176 + retrieval_metrics: text t's target is image t; hit when t is among the 10 / 50 most
    similar images."""
    img = np.array(image_embeddings); img = img / np.linalg.norm(img, axis=1, keepdims=True)
    txt = np.array(text_embeddings); txt = txt / np.linalg.norm(txt, axis=1, keepdims=True)
    p10 = p50 = 0
    for t, tb in enumerate(txt):
        best = tb.dot(img.T).argsort()[-50:][::-1]
        p10 += int(t in best[:10])
        p50 += int(t in best[:50])
    return {"p@10": p10 / len(img), "p@50": p50 / len(img)}


# --------------------------------------------------------------------------
# BERT uncased WordPiece (third-party `tokenizers` / transformers BertTokenizer; published algorithm: BasicTokenizer
# lower-casing + accent stripping + punctuation splitting, then greedy longest-match-first WordPiece with "##"
# continuation pieces and [UNK] for words with no segmentation).  Pins the CALL the reference makes
# (keep_inference.py:99, utils.py:73), not the PubMedBERT vocabulary, which is not available offline.
# --------------------------------------------------------------------------
def wordpiece_ids(vocab: Mapping[str, int], text: str, max_length: int = 256) -> Tuple[List[int], List[int]]:
    import unicodedata

    def is_punct(ch):
        cp = ord(ch)
        return (33 <= cp <= 47) or (58 <= cp <= 64) or (91 <= cp <= 96) or (123 <= cp <= 126) or unicodedata.category(ch).startswith("P")

    text = unicodedata.normalize("NFD", text.lower())
    text = "".join(c for c in text if unicodedata.category(c) != "Mn")
    words: List[str] = []
    for w in text.split():
        cur = ""
        for ch in w:
            if is_punct(ch):
                if cur:
                    words.append(cur)
                words.append(ch)
                cur = ""
            else:
                cur += ch
        if cur:
            words.append(cur)
    pieces: List[int] = []
    for w in words:
        start, sub = 0, []
        while start < len(w):
            end = len(w)
            piece = None
            while start < end:
                cand = ("##" if start else "") + w[start:end]
                if cand in vocab:
                    piece = cand
                    break
                end -= 1
            if piece is None:
                sub = [vocab["[UNK]"]]
                break
            sub.append(vocab[piece])
            start = end
        pieces.extend(sub)
    pieces = pieces[: max_length - 2]
    ids = [vocab["[CLS]"]] + pieces + [vocab["[SEP]"]]
    mask = [1] * len(ids) + [0] * (max_length - len(ids))
    ids = ids + [vocab["[PAD]"]] * (max_length - len(ids))
    return ids, mask

"""Tile-level zero-shot evaluation protocol on the GPU.

Mirrors ``training/path_training/zero_shot.py:91-232`` (``zero_shot_eval``): 50 prompt rounds of zero-shot
classification scored by weighted F1 (median / Q1 / Q3 over the rounds) and text->image retrieval p@10 / p@50, with
the metric definitions of ``training/path_open_clip/zeroshot_metrics.py``.  The reference pulls every embedding to
numpy and loops in Python (50 GEMMs + N Python argmaxes per round; one dot product + argsort per caption); here the
embeddings stay on the device and the two loops are ``keep_group_argmax`` (one fp32 GEMM ``[N,768]x[768,50*C]`` +
per-(tile, round) argmax) and ``keep_retrieval_rank`` (fp32 GEMM + rank of the target per caption).  Only the
``[N,50]`` int32 labels / ``[P]`` int32 ranks cross to the host, where the F1 arithmetic runs in float64 as sklearn's.
"""
from __future__ import annotations

import json
from typing import Dict, Iterable, List, Mapping, Optional, Sequence, Union

import numpy as np
import torch

from . import _lib
from .model import _ptr, _stream
from .wsi import _engine, _normalized

ROUNDS = 50          # zero_shot.py:57 / :125 hard-code 50 prompt rounds


def label2cap(prompts: Union[str, Mapping]) -> Dict[str, List[str]]:
    """zero_shot.py:49-62.  ``prompts``: the parsed prompt JSON or its path (``cfg.DATASET.ZEROSHOT_CLS_PROMPTS``)."""
    if isinstance(prompts, str):
        with open(prompts) as f:
            prompts = json.load(f)
    label_captions: Dict[str, List[str]] = {}
    for type_name in list(prompts["0"]["classnames"].keys()):
        label_captions[type_name] = [
            prompts[str(i)]["templates"].replace("CLASSNAME", prompts[str(i)]["classnames"][type_name]) for i in range(ROUNDS)]
    return label_captions


def weighted_f1_from_confusion(conf: np.ndarray) -> float:
    """sklearn ``f1_score(average='weighted')`` (zeroshot_metrics.py:31) from a confusion matrix ``conf[true, pred]``:
    per-class F1 (0 where undefined) weighted by the class's true count."""
    conf = conf.astype(np.float64)
    tp = np.diag(conf)
    fp = conf.sum(0) - tp
    fn = conf.sum(1) - tp
    den = 2 * tp + fp + fn
    f = np.divide(2 * tp, den, out=np.zeros_like(den), where=den > 0)
    return float(np.average(f, weights=tp + fn))


def classification_rounds(model, image_embeddings: torch.Tensor, cap_embeddings: Mapping[str, torch.Tensor],
                          label_list: Sequence) -> np.ndarray:
    """zero_shot.py:118-139 -> WF1 of each of the 50 prompt rounds (float64 [50]).

    ``cap_embeddings[type_name]`` is ``[50, D]`` (row i = the class's caption of round i); ``label_list`` holds the
    true class names (any label outside the caption classes counts as never predicted, as in sklearn)."""
    m = _engine(model)
    names = list(cap_embeddings.keys())
    C = len(names)
    img = _normalized(m, torch.as_tensor(image_embeddings))
    caps = torch.stack([torch.as_tensor(cap_embeddings[n]).to(m._device, torch.float32) for n in names])      # [C, 50, D]
    if caps.shape[1] < ROUNDS:
        raise ValueError(f"need {ROUNDS} caption embeddings per class, got {caps.shape[1]}")
    K, D = ROUNDS, caps.shape[2]
    bank = _normalized(m, caps[:, :K].permute(1, 0, 2).reshape(K * C, D))                                      # row k*C + c
    N = img.shape[0]
    labels = torch.empty((N, K), dtype=torch.int32, device=m._device)
    rc = _lib.load().keep_group_argmax(m._handle, _ptr(img), _ptr(bank), N, K, C, D, _ptr(labels), _stream(m._device))
    _lib.check(m._handle, rc, "group_argmax")
    pred = labels.cpu().numpy()
    all_names = names + sorted(set(label_list) - set(names))
    index = {n: i for i, n in enumerate(all_names)}
    true = np.array([index[t] for t in label_list], dtype=np.int64)
    if true.shape[0] != N:
        raise ValueError(f"{N} embeddings but {true.shape[0]} labels")
    A = len(all_names)
    out = np.empty(K)
    for k in range(K):
        conf = np.bincount(true * A + pred[:, k], minlength=A * A).reshape(A, A)
        out[k] = weighted_f1_from_confusion(conf)
    return out


def retrieval_ranks(model, image_embeddings: torch.Tensor, text_embeddings: torch.Tensor,
                    targets: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Position of each caption's target image in its descending similarity list (int32 [P] on the device)."""
    m = _engine(model)
    img = _normalized(m, torch.as_tensor(image_embeddings))
    txt = _normalized(m, torch.as_tensor(text_embeddings))
    P, D = txt.shape
    tgt = None if targets is None else torch.as_tensor(targets).to(m._device, torch.int32).contiguous()
    if tgt is not None and (tgt.numel() != P or int(tgt.min()) < 0 or int(tgt.max()) >= img.shape[0]):
        raise ValueError("targets must be [P] image indices")
    rank = torch.empty(P, dtype=torch.int32, device=m._device)
    rc = _lib.load().keep_retrieval_rank(m._handle, _ptr(txt), _ptr(img), P, img.shape[0], D, _ptr(tgt), _ptr(rank), _stream(m._device))
    _lib.check(m._handle, rc, "retrieval_rank")
    return rank


def retrieval_metrics(model, image_embeddings: torch.Tensor, text_embeddings: torch.Tensor) -> Dict[str, float]:
    """zero_shot.py:161-176 + zeroshot_metrics.py:6-17 (caption t must retrieve image t)."""
    rank = retrieval_ranks(model, image_embeddings, text_embeddings).cpu().numpy()
    n = image_embeddings.shape[0]                # the reference divides by len(y_target) = number of images
    return {"p@10": int((rank < 10).sum()) / n, "p@50": int((rank < 50).sum()) / n}


def _batches(split) -> Iterable:
    return split.dataloader if hasattr(split, "dataloader") else split


def _tokenize(tokenizer, texts: Sequence[str], device):
    # zero_shot.py:72 (`pad_to_max_length=True` is the deprecated spelling of padding='max_length')
    enc = tokenizer(list(texts), add_special_tokens=True, max_length=256, padding="max_length", truncation=True, return_tensors="pt")
    return enc.to(device) if hasattr(enc, "to") else {k: v.to(device) for k, v in enc.items()}


def zero_shot_eval(model, tokenizer, data: Mapping, prompts: Union[str, Mapping, None] = None) -> Dict[str, float]:
    """zero_shot.py:79-232 for a :class:`keep_amd.KEEPModel`.

    ``data`` may hold ``'zeroshot_cls'`` (batches of ``(images, labels)``), ``'zeroshot_ret'`` and ``'zeroshot_po'``
    (batches of ``(images, captions)``); each value is an iterable of batches or an object with ``.dataloader`` as in
    the reference.  Returns the reference's result keys."""
    m = _engine(model)
    dev = m._device
    results: Dict[str, float] = {}
    if "zeroshot_cls" in data:
        feats, label_list = [], []
        for images, labels in _batches(data["zeroshot_cls"]):
            feats.append(m.encode_image(images.to(dev)))
            label_list.extend(labels)
        caps = {name: m.encode_text(_tokenize(tokenizer, c, dev)) for name, c in label2cap(prompts).items()}
        val_cls = classification_rounds(m, torch.cat(feats), caps, label_list)
        q1, med, q3 = np.percentile(val_cls, (25, 50, 75), method="midpoint")           # zero_shot.py:217
        results["zeroshot-cls-WF1-median"], results["zeroshot-cls-WF1-Q1"], results["zeroshot-cls-WF1-Q3"] = float(med), float(q1), float(q3)
    for key, tag in (("zeroshot_ret", "ret"), ("zeroshot_po", "po")):
        if key in data:
            fi, ft = [], []
            for images, texts in _batches(data[key]):
                out = m(images.to(dev), _tokenize(tokenizer, texts, dev))
                fi.append(out["vision_features"]); ft.append(out["text_features"])
            r = retrieval_metrics(m, torch.cat(fi), torch.cat(ft))
            results[f"zeroshot-{tag}-p@10"], results[f"zeroshot-{tag}-p@50"] = r["p@10"], r["p@50"]
    return results

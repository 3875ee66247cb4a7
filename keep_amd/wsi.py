"""GPU versions of the reference's slide-level zero-shot logic.

Function names, arguments and return values mirror ``WSI_evaluation/utils.py``,
``subtyping_utils.py``, ``detection_utils.py`` and ``segment_utils.py`` so the three
``zeroshot_*_WSI.py`` scripts can import them instead; the per-classifier / per-tile Python loops of
the reference (one GEMM + ``.item()`` sync per prompt set, one ``.cpu()`` per tile) become a handful of
kernel launches through the C ABI (``keep_similarity``, ``keep_prompt_scores``, ``keep_refine``).

``model`` is a :class:`keep_amd.KEEPModel`; where the reference passes ``KEEP_model`` (a dict with
``'model'`` and ``'tokenizer'``) the same dict is accepted.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Mapping, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _lib
from .model import KEEPModel, _ptr, _stream


# ------------------------------------------------------------------------------------------------
# classifier construction (utils.py:64-104) with a prompt-string cache
# ------------------------------------------------------------------------------------------------
class TextEmbeddingCache:
    """The RCC prompt bank asks for 7128 ``encode_text`` calls but holds only 264 distinct strings
    (SURVEY.md §3.2): embed each distinct string once, in batches."""

    def __init__(self, KEEP_model: Mapping, device, batch: int = 64, max_length: int = 256):
        self.model, self.tokenizer = KEEP_model["model"], KEEP_model["tokenizer"]
        self.device, self.batch, self.max_length = device, batch, max_length
        self._cache: Dict[str, torch.Tensor] = {}

    def embed(self, texts: Sequence[str]) -> torch.Tensor:
        todo = [t for t in dict.fromkeys(texts) if t not in self._cache]
        for i in range(0, len(todo), self.batch):
            chunk = todo[i:i + self.batch]
            # same tokenizer call as utils.py:73
            enc = self.tokenizer(chunk, max_length=self.max_length, padding="max_length", truncation=True, return_tensors="pt")
            enc = enc.to(self.device) if hasattr(enc, "to") else {k: v.to(self.device) for k, v in enc.items()}
            emb = self.model.encode_text(enc)
            for t, e in zip(chunk, emb):
                self._cache[t] = e
        return torch.stack([self._cache[t] for t in texts])


def zero_shot_classifier(KEEP_model, classnames, templates, device, cache: Optional[TextEmbeddingCache] = None):
    """utils.py:64-84.  Returns [feat_dim, num_classes]."""
    cache = cache or TextEmbeddingCache(KEEP_model, device)
    weights = []
    for classname in classnames:
        if isinstance(templates, list):
            texts = [t.replace("CLASSNAME", classname) for t in templates]
        else:
            texts = [templates.replace("CLASSNAME", classname)]
        # the reference keeps only row 0 of the batch (`encode_text(text_inputs)[0]`, utils.py:74)
        class_embeddings = cache.embed(texts)[0].unsqueeze(0)
        class_embedding = torch.nn.functional.normalize(class_embeddings, dim=-1).mean(dim=0)
        class_embedding = class_embedding / class_embedding.norm()
        weights.append(class_embedding)
    return torch.stack(weights, dim=1).to(device)


def get_zeroshot_classifier(model, label_map, prompts, device, add_normal=False, cache: Optional[TextEmbeddingCache] = None):
    """utils.py:86-104."""
    classnames, templates = prompts["classnames"], prompts["templates"]
    idx_to_class = {v: k for k, v in label_map.items()}
    n_classes = len(idx_to_class)
    if add_normal:
        idx_to_class[n_classes] = "Normal"
        n_classes = len(idx_to_class)
    classnames_text = [classnames[idx_to_class[idx]] for idx in range(n_classes)]
    return zero_shot_classifier(model, classnames_text, templates, device, cache)


def build_classifier_bank(KEEP_model, label_map, prompts, device, add_normal=False,
                          cache: Optional[TextEmbeddingCache] = None) -> List[torch.Tensor]:
    """All prompt classifiers of a prompt file at once: the loop the three scripts run over ``prompts[str(i)]``
    (zeroshot_subtyping_WSI.py:59-63 -> ``get_zeroshot_classifier`` per prompt set) with every distinct string embedded
    once and the per-class normalise -> mean -> renormalise (utils.py:76-80) done for all K x C columns in three tensor
    ops instead of K x C x 4 tiny launches.  ``prompts``: the parsed prompt JSON ({"0": {"classnames", "templates"}, ...})
    or a list of such entries.  Returns K tensors [feat_dim, C] (views of one [K, feat_dim, C] tensor)."""
    cache = cache or TextEmbeddingCache(KEEP_model, device)
    entries = [prompts[str(i)] for i in range(len(prompts))] if isinstance(prompts, Mapping) else list(prompts)
    idx_to_class = {v: k for k, v in label_map.items()}
    if add_normal:
        idx_to_class[len(idx_to_class)] = "Normal"
    C_ = len(idx_to_class)
    texts = []
    for e in entries:
        tpl = e["templates"][0] if isinstance(e["templates"], list) else e["templates"]     # row 0 only, as utils.py:74
        texts.extend(tpl.replace("CLASSNAME", e["classnames"][idx_to_class[c]]) for c in range(C_))
    emb = cache.embed(texts).to(device, torch.float32)                                      # [K*C, D]
    emb = torch.nn.functional.normalize(emb, dim=-1)           # normalize(class_embeddings, dim=-1).mean(dim=0) over ONE row
    emb = emb / emb.norm(dim=-1, keepdim=True)                 # class_embedding /= class_embedding.norm()
    bank = emb.reshape(len(entries), C_, -1).permute(0, 2, 1).contiguous()                  # [K, D, C]
    return list(bank.unbind(0))


# ------------------------------------------------------------------------------------------------
def _engine(model) -> KEEPModel:
    m = model["model"] if isinstance(model, Mapping) else model
    if not isinstance(m, KEEPModel):
        raise TypeError("expected a keep_amd.KEEPModel (or the reference's KEEP_model dict holding one)")
    m._ready_device()
    return m


def _normalized(m: KEEPModel, feats: torch.Tensor) -> torch.Tensor:
    f = feats.to(m._device, torch.float32)
    if f.dim() == 3:
        f = f.squeeze(0)
    f = f.contiguous().clone()
    _lib.check(m._handle, _lib.load().keep_op_l2norm(m._handle, _ptr(f), f.shape[0], f.shape[1], _stream(m._device)), "l2norm")
    return f


def rank_cls_score(model, logits: torch.Tensor) -> float:
    """utils.py:107-117 on a ready [N,C] logits matrix."""
    m = _engine(model)
    x = logits.to(m._device, torch.float32).contiguous()
    eye = torch.eye(x.shape[1], device=m._device)
    # scores of ONE classifier whose logits are given: run the group kernel with K = 1 on logits @ I
    return float(prompt_scores(m, x, [eye], pre_normalized=True, _logits_given=True)[0])


def prompt_scores(model, tile_features: torch.Tensor, classifiers: Sequence[torch.Tensor], pre_normalized: bool = False,
                  _logits_given: bool = False) -> torch.Tensor:
    """rank_cls_score of every classifier in one pass -> fp32 [K] on the engine's device."""
    m = _engine(model)
    K, (D, Cc) = len(classifiers), classifiers[0].shape
    bank = torch.stack([c.to(m._device, torch.float32).t() for c in classifiers]).reshape(K * Cc, D).contiguous()
    f = tile_features.to(m._device, torch.float32).contiguous() if pre_normalized else _normalized(m, tile_features)
    if _logits_given:       # f already holds logits [N, C]; bank is the identity: reuse the same kernels
        D = f.shape[1]
        if D % 16:
            pad = 16 - D % 16
            f = torch.nn.functional.pad(f, (0, pad)).contiguous()
            bank = torch.nn.functional.pad(bank, (0, pad)).contiguous()
            D += pad
    scores = torch.empty(K, dtype=torch.float32, device=m._device)
    rc = _lib.load().keep_prompt_scores(m._handle, _ptr(f), _ptr(bank), f.shape[0], K, Cc, D, _ptr(scores), _stream(m._device))
    _lib.check(m._handle, rc, "prompt_scores")
    return scores


def zero_shot_prompt_select(model, classifiers, tile_features, topn, device=None):
    """utils.py:119-146: score every prompt classifier on the slide, keep the top-n, sum and renormalise.

    ``model`` is an extra leading argument compared with the reference (which only used torch ops)."""
    m = _engine(model)
    scores = prompt_scores(m, tile_features, classifiers)
    # same tie behaviour as the reference: torch.sort(descending=True) on a CPU tensor of Python floats
    _, index = torch.sort(torch.tensor(scores.cpu().tolist()), descending=True)
    merge = torch.zeros_like(classifiers[0].to(m._device, torch.float32))
    for cls_index in index[0:topn]:
        merge += classifiers[int(cls_index)].to(m._device, torch.float32)
    return torch.nn.functional.normalize(merge, p=2, dim=0)


def random_prompt_ensemble(classifiers: Sequence[torch.Tensor], topn: int) -> torch.Tensor:
    """The `prompt_screening = False` branch of the three scripts (zeroshot_subtyping_WSI.py:68-76): `topn` picks with
    `random.seed(c); random.randint(0, K-1)` for c = 0..topn-1 (so the picks are the same on every run), summed and
    column-normalised."""
    import random
    ensemble_cls = torch.zeros_like(classifiers[-1])
    for cter in range(topn):
        random.seed(cter)
        ensemble_cls += classifiers[random.randint(0, len(classifiers) - 1)]
    return torch.nn.functional.normalize(ensemble_cls, p=2, dim=0)


# ------------------------------------------------------------------------------------------------
def _probs(m: KEEPModel, classifier: torch.Tensor, tile_features: torch.Tensor) -> torch.Tensor:
    """softmax(10 * normalize(feat) @ classifier, dim=1) -- subtyping_utils.py:69-72."""
    f = _normalized(m, tile_features)
    return m.similarity(f, classifier.t().contiguous(), scale=10.0, mode="softmax")


def refine(model, probs: torch.Tensor, tile_coords, patch_size: int, overlap: bool) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Core of the three ``refine_seg`` variants.  Returns (coords [U,2], mean probs [U,C], index [U]) for the
    U distinct coordinates in first-seen order (the key order of the reference's dicts)."""
    m = _engine(model)
    p = probs.to(m._device, torch.float32).contiguous()
    coords = torch.as_tensor(np.asarray(tile_coords)).to(torch.int64)
    lim = 2 ** 31 - 1 - abs(int(patch_size))
    if coords.numel() and (int(coords.min()) < -lim or int(coords.max()) > lim):
        raise ValueError("tile coordinates must fit int32 (the device-side coordinate hash packs (x, y) into 64 bits)")
    coords = coords.to(m._device).contiguous()
    N, Cc = p.shape
    out = torch.empty_like(p)
    first = torch.empty(N, dtype=torch.int32, device=m._device)
    rc = _lib.load().keep_refine(m._handle, _ptr(p), _ptr(coords), N, Cc, int(patch_size), int(bool(overlap)), _ptr(out), _ptr(first),
                                 _stream(m._device))
    _lib.check(m._handle, rc, "refine")
    idx = torch.nonzero(first, as_tuple=False).squeeze(1)
    return coords[idx], out[idx], idx


def zero_shot_subtyping(model, classifier, tile_features, tile_coords, patch_size=256, overlap=True):
    """subtyping_utils.py:67-83 -> slide label (int tensor, like the reference's ``max_label``)."""
    m = _engine(model)
    _, mean, _ = refine(m, _probs(m, classifier, tile_features), tile_coords, patch_size, overlap)
    pred = mean.argmax(dim=1)
    C_ = classifier.shape[1]
    # (preds == ix).sum() / len(preds) in float64, as the numpy expression at subtyping_utils.py:80
    frac = [float((pred == ix).sum().item()) / pred.shape[0] for ix in range(C_)]
    _, max_label = torch.tensor(frac[0:-1]).max(0)
    return max_label


def zero_shot_detection(model, classifier, tile_features, tile_coords, patch_size=256, overlap=False, threshold=0.5):
    """detection_utils.py:88-100 -> tumour-tile ratio."""
    m = _engine(model)
    _, mean, _ = refine(m, _probs(m, classifier, tile_features), tile_coords, patch_size, overlap)
    return float((mean[:, 1] > threshold).sum().item()) / mean.shape[0]


def zero_shot_segment_probs(model, classifier, tile_features, tile_coords, patch_size=224, overlap=True) -> Dict[str, float]:
    """segment_utils.py:44-52 + refine_seg :63-89 -> {"x_y": tumour probability} in first-seen order
    (the AUC / Dice evaluation against openslide masks that follows in the reference is out of scope)."""
    m = _engine(model)
    coords, mean, _ = refine(m, _probs(m, classifier, tile_features), tile_coords, patch_size, overlap)
    c, v = coords.cpu().tolist(), mean[:, 1].cpu().tolist()
    return {f"{x}_{y}": p for (x, y), p in zip(c, v)}

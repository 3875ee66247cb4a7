"""GPU versions of the reference's slide-level zero-shot logic.

Every public function has the NAME, POSITIONAL ARGUMENTS, DEFAULTS and RETURN VALUE of its counterpart in
``WSI_evaluation/utils.py``, ``subtyping_utils.py``, ``detection_utils.py`` and ``segment_utils.py`` (cited per function), so
the call sites of the three ``zeroshot_*_WSI.py`` scripts run unchanged (``keep_amd/wsi_evaluation/`` re-exports them under the
reference's module names).  The per-classifier / per-tile Python loops of the reference (one GEMM + ``.item()`` sync per prompt
set, one ``.cpu()`` per tile) become a handful of kernel launches through the C ABI (``keep_similarity``, ``keep_prompt_scores``,
``keep_refine``).

The reference's functions are plain torch code and take no model; here the kernels belong to an engine handle, which is found
from the device (``keep_amd.model.engine_for``: the live ``KEEPModel`` on that GPU, or a weight-less handle).  An optional
keyword ``model=`` pins it (a ``KEEPModel`` or the reference's ``KEEP_model`` dict).
"""
from __future__ import annotations

import weakref
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .model import KEEPModel, _ptr, _stream, engine_for


# ------------------------------------------------------------------------------------------------
# classifier construction (utils.py:64-104) with a prompt-string cache
# ------------------------------------------------------------------------------------------------
class TextEmbeddingCache:
    """The RCC prompt bank asks for 7128 ``encode_text`` calls but holds only 264 distinct strings
    (SURVEY.md §3.2): embed each distinct string once, in batches.  A cache that belongs to an engine (``_cache_for``) refers to it
    weakly, so that it never keeps the model -- and its GPU weights -- alive."""

    def __init__(self, KEEP_model: Mapping, device, batch: int = 64, max_length: int = 256, weak: bool = False):
        model = KEEP_model["model"]
        self._model = weakref.ref(model) if weak else (lambda: model)
        self.tokenizer = KEEP_model["tokenizer"]
        self.device, self.batch, self.max_length = device, batch, max_length
        self._cache: Dict[str, torch.Tensor] = {}

    @property
    def model(self):
        m = self._model()
        if m is None:
            raise ReferenceError("the model this prompt cache was built for no longer exists")
        return m

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


# one cache per (model, tokenizer, weights, precision setting): the reference scripts call get_zeroshot_classifier once per prompt set with
# the same KEEP_model dict (zeroshot_subtyping_WSI.py:59-62), so repeated strings are embedded once without any change to the call site.
# The cache lives ON the model object (and refers back to it weakly): it goes away with the model, and nothing global pins a model.
PROMPT_CACHE = True          # False: embed every string on every call, exactly as the reference does
TEXT_OPTIONS = ("precision", "strict_blocks")      # the engine options encode_text depends on


def _cache_for(KEEP_model, device) -> TextEmbeddingCache:
    m, tok = KEEP_model["model"], KEEP_model["tokenizer"]
    if not PROMPT_CACHE or not isinstance(m, KEEPModel):
        return TextEmbeddingCache(KEEP_model, device)
    # embeddings depend on the weights and on the precision setting of the TEXT tower (precision, strict_blocks): set_precision after the first
    # call must not return embeddings computed under the old setting.  Image-only options (the per-block plan, lanes, kernel selection) do not
    # touch the text tower and leave the cache alone.
    key = (id(tok), getattr(m, "_weights_epoch", 0), str(device), tuple((k, float(m._options.get(k, 0))) for k in TEXT_OPTIONS))
    slot = getattr(m, "_prompt_cache", None)
    if slot is None or slot[0] != key:
        slot = (key, TextEmbeddingCache(KEEP_model, device, weak=True))
        m._prompt_cache = slot
    return slot[1]


def _unit_rows(m: KEEPModel, x: torch.Tensor) -> torch.Tensor:
    """Rows of a 2-D tensor divided by max(||row||, 1e-12) on the engine (``F.normalize(x, dim=-1)`` semantics; keep_op_l2norm) --
    torch holds the storage, the arithmetic is the engine's fp32 row kernel; the result comes back in the dtype and on the device of ``x``
    (the reference's ``F.normalize`` keeps both).  There is deliberately no torch fallback: with a text encoder that is not a ``KEEPModel``
    the normalisation still runs on the engine of the device (a weight-less handle: these kernels need no weights), and without a GPU the
    call raises like every other entry point of this package."""
    dev, dt = x.device, x.dtype
    r = x.detach().to(m._device, torch.float32).contiguous().clone()
    if r.numel():
        _lib.check(m._handle, _lib.load().keep_op_l2norm(m._handle, _ptr(r), r.shape[0], r.shape[1], _stream(m._device)), "l2norm")
    if dt.is_floating_point and dt != torch.float32:
        r = r.to(dt)
    return r if dev == m._device else r.to(dev)


def zero_shot_classifier(KEEP_model, classnames, templates, device, cache: Optional[TextEmbeddingCache] = None):
    """utils.py:64-84.  Returns [feat_dim, num_classes]."""
    cache = cache or _cache_for(KEEP_model, device)
    eng = _engine(KEEP_model["model"] if isinstance(KEEP_model["model"], KEEPModel) else None, device=device)
    weights = []
    with torch.no_grad():
        for classname in classnames:
            if isinstance(templates, list):
                texts = [t.replace("CLASSNAME", classname) for t in templates]
            else:
                texts = [templates.replace("CLASSNAME", classname)]
            # the reference keeps only row 0 of the batch (`encode_text(text_inputs)[0]`, utils.py:74)
            class_embeddings = cache.embed(texts)[0].unsqueeze(0)
            # F.normalize(class_embeddings, dim=-1).mean(dim=0) over ONE row, then / norm (utils.py:76-80): two passes of the row kernel
            class_embedding = _unit_rows(eng, _unit_rows(eng, class_embeddings))[0]
            weights.append(class_embedding)
    return torch.stack(weights, dim=1).to(device)


def get_zeroshot_classifier(model, label_map, prompts, device, add_normal=False, cache: Optional[TextEmbeddingCache] = None):
    """utils.py:86-104 (``model`` is the reference's ``KEEP_model`` dict: {'model', 'tokenizer', ...})."""
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
    cache = cache or _cache_for(KEEP_model, device)
    entries = [prompts[str(i)] for i in range(len(prompts))] if isinstance(prompts, Mapping) else list(prompts)
    idx_to_class = {v: k for k, v in label_map.items()}
    if add_normal:
        idx_to_class[len(idx_to_class)] = "Normal"
    C_ = len(idx_to_class)
    texts = []
    for e in entries:
        tpl = e["templates"][0] if isinstance(e["templates"], list) else e["templates"]     # row 0 only, as utils.py:74
        texts.extend(tpl.replace("CLASSNAME", e["classnames"][idx_to_class[c]]) for c in range(C_))
    eng = _engine(KEEP_model["model"] if isinstance(KEEP_model["model"], KEEPModel) else None, device=device)
    emb = cache.embed(texts).to(device, torch.float32)                                      # [K*C, D]
    emb = _unit_rows(eng, emb)                                 # normalize(class_embeddings, dim=-1).mean(dim=0) over ONE row
    emb = _unit_rows(eng, emb)                                 # class_embedding /= class_embedding.norm()
    bank = emb.reshape(len(entries), C_, -1).permute(0, 2, 1).contiguous()                  # [K, D, C]
    return list(bank.unbind(0))


# ------------------------------------------------------------------------------------------------
def _engine(model=None, *tensors, device=None) -> KEEPModel:
    """The engine of a call: ``model`` if pinned, else the one that lives on the device of ``device`` / the first GPU tensor."""
    if model is not None:
        return engine_for(model=model)
    if device is None:
        for t in tensors:
            if isinstance(t, torch.Tensor) and t.device.type == "cuda":
                device = t.device
                break
    return engine_for(device=device)


def _normalized(m: KEEPModel, feats: torch.Tensor) -> torch.Tensor:
    f = feats.to(m._device, torch.float32)
    if f.dim() == 3:
        f = f.squeeze(0)
    f = f.contiguous().clone()
    _lib.check(m._handle, _lib.load().keep_op_l2norm(m._handle, _ptr(f), f.shape[0], f.shape[1], _stream(m._device)), "l2norm")
    return f


def rank_cls_score(logits: torch.Tensor, model=None) -> float:
    """utils.py:107-117 on a ready [N,C] logits matrix -> Python float."""
    m = _engine(model, logits)
    x = logits.to(m._device, torch.float32).contiguous()
    eye = torch.eye(x.shape[1], device=m._device)
    # scores of ONE classifier whose logits are given: run the group kernel with K = 1 on logits @ I
    return float(prompt_scores(x, [eye], pre_normalized=True, model=m, _logits_given=True)[0])


def prompt_scores(tile_features: torch.Tensor, classifiers: Sequence[torch.Tensor], pre_normalized: bool = False, model=None,
                  _logits_given: bool = False) -> torch.Tensor:
    """rank_cls_score of every classifier in one pass (the loop at utils.py:127-130) -> fp32 [K] on the engine's device."""
    m = _engine(model, tile_features, *classifiers[:1])
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


def zero_shot_prompt_select(classifiers, tile_features, topn, device, model=None):
    """utils.py:119-146: score every prompt classifier on the slide, keep the top-n, sum and renormalise -> [feat_dim, C] on
    the device of the classifiers (``torch.zeros_like(classifiers[0])``, utils.py:141)."""
    m = _engine(model, device=device)
    scores = prompt_scores(tile_features, classifiers, model=m)
    # same tie behaviour as the reference: torch.sort(descending=True) on a CPU tensor of Python floats (utils.py:139)
    _, index = torch.sort(torch.tensor(scores.cpu().tolist()), descending=True)
    merge = torch.zeros_like(classifiers[0], dtype=torch.float32)
    for cls_index in index[0:topn]:
        merge += classifiers[int(cls_index)].to(merge.device, torch.float32)
    return _unit_rows(m, merge.t()).t().contiguous()            # F.normalize(merge, p=2, dim=0): the columns are the class vectors


def random_prompt_ensemble(classifiers: Sequence[torch.Tensor], topn: int, model=None) -> torch.Tensor:
    """The `prompt_screening = False` branch of the three scripts (zeroshot_subtyping_WSI.py:68-76): `topn` picks with
    `random.seed(c); random.randint(0, K-1)` for c = 0..topn-1 (so the picks are the same on every run), summed and
    column-normalised."""
    import random
    ensemble_cls = torch.zeros_like(classifiers[-1])
    for cter in range(topn):
        random.seed(cter)
        ensemble_cls += classifiers[random.randint(0, len(classifiers) - 1)]
    return _unit_rows(_engine(model, ensemble_cls), ensemble_cls.t()).t().contiguous()


def cood2str(cood):
    """utils.py:148-149."""
    return str(cood[0]) + "_" + str(cood[1])


def str2cood(str):                      # noqa: A002 -- the reference's parameter name (utils.py:150)
    """utils.py:150-151."""
    return [int(item) for item in str.split("_")]


def accuracy(logits, target, topk=(1,)):
    """utils.py:153-156."""
    pred = logits.topk(max(topk), 1, True, True)[1].t()
    correct = pred.eq(target.view(1, -1).expand_as(pred))
    return [float(correct[:k].reshape(-1).float().sum(0, keepdim=True).cpu().numpy()) for k in topk]


# ------------------------------------------------------------------------------------------------
def _probs(m: KEEPModel, classifier: torch.Tensor, tile_features: torch.Tensor) -> torch.Tensor:
    """softmax(10 * normalize(feat) @ classifier, dim=1) -- subtyping_utils.py:69-72."""
    f = _normalized(m, tile_features)
    return m.similarity(f, classifier.t().contiguous(), scale=10.0, mode="softmax")


def refine(probs: torch.Tensor, tile_coords, patch_size: int, overlap: bool, model=None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Core of the three ``refine_seg`` variants.  Returns (coords [U,2], mean probs [U,C], index [U]) for the
    U distinct coordinates in first-seen order (the key order of the reference's dicts)."""
    m = _engine(model, probs)
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


def _keys(coords: torch.Tensor) -> List[str]:
    return [f"{x}_{y}" for x, y in coords.cpu().tolist()]


def refine_seg_subtyping(logits_slide, coords_slide, patch_size=224, overlap=True, model=None) -> Dict[str, int]:
    """``refine_seg`` of subtyping_utils.py:38-65: {"x_y": predicted class} in first-seen order."""
    coords, mean, _ = refine(logits_slide, coords_slide, patch_size, overlap, model=model)
    return dict(zip(_keys(coords), mean.argmax(dim=1).cpu().tolist()))


def zero_shot_subtyping(classifier, tile_features, tile_coords, patch_size=256, overlap=True, model=None):
    """subtyping_utils.py:67-83 -> slide label (0-d int64 tensor, the reference's ``max_label``)."""
    m = _engine(model, tile_features, classifier)
    _, mean, _ = refine(_probs(m, classifier, tile_features), tile_coords, patch_size, overlap, model=m)
    pred = mean.argmax(dim=1)
    C_ = classifier.shape[1]
    # (preds == ix).sum() / len(preds) in float64, as the numpy expression at subtyping_utils.py:80
    frac = [float((pred == ix).sum().item()) / pred.shape[0] for ix in range(C_)]
    _, max_label = torch.tensor(frac[0:-1]).max(0)
    return max_label


def refine_seg_detection(logits_slide, coords_slide, patch_size=224, threshold=0.5, overlap=True, model=None):
    """``refine_seg`` of detection_utils.py:39-74: ({"x_y": 0/1}, {"x_y": tumour probability})."""
    coords, mean, _ = refine(logits_slide, coords_slide, patch_size, overlap, model=model)
    keys, p1 = _keys(coords), mean[:, 1]
    return dict(zip(keys, (p1 > threshold).to(torch.int64).cpu().tolist())), dict(zip(keys, p1.cpu().tolist()))


def zero_shot_detection(classifier, tile_features, tile_coords, patch_size=256, overlap=False, model=None):
    """detection_utils.py:88-100 -> tumour-tile ratio (float)."""
    m = _engine(model, tile_features, classifier)
    _, mean, _ = refine(_probs(m, classifier, tile_features), tile_coords, patch_size, overlap, model=m)
    return float((mean[:, 1] > 0.5).sum().item()) / mean.shape[0]


def refine_seg_segment(logits_slide, coords_slide, patch_size=224, overlap=True, model=None) -> Dict[str, float]:
    """``refine_seg`` of segment_utils.py:63-89: {"x_y": tumour probability} in first-seen order."""
    coords, mean, _ = refine(logits_slide, coords_slide, patch_size, overlap, model=model)
    return dict(zip(_keys(coords), mean[:, 1].cpu().tolist()))


def zero_shot_segment_probs(classifier, tile_features, tile_coords, patch_size=224, overlap=True, model=None) -> Dict[str, float]:
    """segment_utils.py:44-52 + refine_seg :63-89: the dense per-tile tumour-probability map of ``zero_shot_segment``."""
    m = _engine(model, tile_features, classifier)
    return refine_seg_segment(_probs(m, classifier, tile_features), tile_coords, patch_size, overlap, model=m)


def zero_shot_segment(classifier, tile_features, tile_coords, mask_path, patch_size=224, overlap=True, model=None):
    """segment_utils.py:44-60.  The probability map is computed here; the AUC / Dice evaluation against an openslide mask that
    follows in the reference (``eval_seg_auc`` / ``eval_seg_coarse``, :91-152) is out of scope (needs
    openslide and real WSI masks) -- pass ``mask_path=None`` to get the map, which is what those two functions consume."""
    probs_all_refined = zero_shot_segment_probs(classifier, tile_features, tile_coords, patch_size, overlap, model=model)
    if mask_path is None:
        return probs_all_refined
    raise NotImplementedError("AUC / Dice against an openslide mask (segment_utils.py:91-152) is outside the hot path; "
                              "call with mask_path=None and feed the returned map to the reference's eval_seg_auc / eval_seg_coarse")

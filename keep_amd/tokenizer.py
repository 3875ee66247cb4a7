"""Tokenizer side of the text path.

The reference tokenises with the HF tokenizer shipped in the release directory (``quick_start/keep_inference.py:87``:
``AutoTokenizer.from_pretrained(model_path)`` -> ``BertTokenizerFast`` over PubMedBERT's uncased WordPiece vocabulary) and
always with the same call (``keep_inference.py:99``, ``WSI_evaluation/utils.py:73``)::

    tokenizer(texts, max_length=256, padding='max_length', truncation=True, return_tensors='pt').to(device)

``WordPieceTokenizer`` is a native implementation of exactly that pipeline -- BERT normalisation (control / whitespace clean-up,
CJK isolation, accent stripping, lower-casing), whitespace + punctuation pre-tokenisation, greedy longest-match WordPiece with
``[UNK]`` for unmatched or over-long words, ``[CLS] .. [SEP]`` framing, truncation and padding -- reading nothing but ``vocab.txt``,
so the engine needs no third-party package at run time.  ``tests/test_tokenizer.py`` pins it token for token against
``transformers.BertTokenizerFast`` (the class the reference gets) on a synthetic vocabulary; PubMedBERT's own vocabulary is not
available offline, so parity on it is unpinned.  ``load_tokenizer(..., backend="hf")`` still opens the directory with the
third-party library instead.
"""
from __future__ import annotations

import os
import unicodedata
from typing import Dict, List, Mapping, Optional, Sequence, Union

import torch

MAX_LENGTH = 256       # keep_inference.py:99 / utils.py:73
SPECIAL_TOKENS = ("[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]")


class Encoding(dict):
    """The ``BatchEncoding`` surface the reference touches: a mapping of tensors with ``.to(device)`` and attribute access."""

    def to(self, device=None, **_):
        return Encoding({k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in self.items()})

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e


def _is_whitespace(ch: str) -> bool:
    # the tokenizers crate tests Rust's char::is_whitespace (Unicode White_Space): the separator categories Zs, Zl (U+2028) and Zp (U+2029);
    # the White_Space members of category Cc (VT, FF, NEL ...) never get here -- the clean-up removes control characters first
    return ch in " \t\n\r" or unicodedata.category(ch) in ("Zs", "Zl", "Zp")


def _is_control(ch: str) -> bool:
    if ch in "\t\n\r":
        return False
    # Cc / Cf / Co as in the crate's `is_other()`; UNASSIGNED code points (Cn) are kept there and end up as [UNK]
    return unicodedata.category(ch) in ("Cc", "Cf", "Co")


def _is_punctuation(ch: str) -> bool:
    cp = ord(ch)
    if (33 <= cp <= 47) or (58 <= cp <= 64) or (91 <= cp <= 96) or (123 <= cp <= 126):
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp: int) -> bool:
    return ((0x4E00 <= cp <= 0x9FFF) or (0x3400 <= cp <= 0x4DBF) or (0x20000 <= cp <= 0x2A6DF) or (0x2A700 <= cp <= 0x2B73F)
            or (0x2B740 <= cp <= 0x2B81F) or (0x2B820 <= cp <= 0x2CEAF) or (0xF900 <= cp <= 0xFAFF) or (0x2F800 <= cp <= 0x2FA1F))


class WordPieceTokenizer:
    """Uncased BERT WordPiece, callable like the HF tokenizer the reference uses."""

    def __init__(self, vocab: Union[str, Mapping[str, int]], do_lower_case: bool = True, max_input_chars_per_word: int = 100):
        if isinstance(vocab, str):
            with open(vocab, encoding="utf-8") as f:
                vocab = {line.rstrip("\n"): i for i, line in enumerate(f)}
        self.vocab: Dict[str, int] = dict(vocab)
        for t in ("[PAD]", "[UNK]", "[CLS]", "[SEP]"):
            if t not in self.vocab:
                raise ValueError(f"vocabulary has no {t} entry")
        self.do_lower_case = do_lower_case
        self.max_chars = max_input_chars_per_word
        self.pad_token_id, self.unk_token_id = self.vocab["[PAD]"], self.vocab["[UNK]"]
        self.cls_token_id, self.sep_token_id = self.vocab["[CLS]"], self.vocab["[SEP]"]
        self._special = [t for t in SPECIAL_TOKENS if t in self.vocab]

    # ---- BertNormalizer: clean text, isolate CJK, strip accents, lower-case
    def _normalize(self, text: str) -> str:
        out = []
        for ch in text:
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or _is_control(ch):
                continue
            if _is_whitespace(ch):
                out.append(" ")
            elif _is_cjk(cp):
                out.extend((" ", ch, " "))
            else:
                out.append(ch)
        text = "".join(out)
        if self.do_lower_case:
            text = "".join(c for c in unicodedata.normalize("NFD", text) if unicodedata.category(c) != "Mn")
            text = text.lower()
        return text

    # ---- BertPreTokenizer: split on whitespace, every punctuation character is a word of its own
    @staticmethod
    def _pre_tokenize(text: str) -> List[str]:
        words, cur = [], []
        for ch in text:
            if _is_whitespace(ch):
                if cur:
                    words.append("".join(cur)); cur = []
            elif _is_punctuation(ch):
                if cur:
                    words.append("".join(cur)); cur = []
                words.append(ch)
            else:
                cur.append(ch)
        if cur:
            words.append("".join(cur))
        return words

    def _wordpiece(self, word: str) -> List[int]:
        if len(word) > self.max_chars:
            return [self.unk_token_id]
        ids, start = [], 0
        while start < len(word):
            end, piece = len(word), None
            while start < end:
                cand = ("##" if start else "") + word[start:end]
                if cand in self.vocab:
                    piece = cand
                    break
                end -= 1
            if piece is None:
                return [self.unk_token_id]
            ids.append(self.vocab[piece])
            start = end
        return ids

    def _split_special(self, text: str) -> List[Union[str, int]]:
        """The fast tokenizer matches its special tokens in the raw text, wherever they occur, before normalisation."""
        parts: List[Union[str, int]] = [text]
        for tok in self._special:
            nxt: List[Union[str, int]] = []
            for p in parts:
                if isinstance(p, int):
                    nxt.append(p)
                    continue
                segs = p.split(tok)
                for i, sgm in enumerate(segs):
                    if i:
                        nxt.append(self.vocab[tok])
                    if sgm:
                        nxt.append(sgm)
            parts = nxt
        return parts

    def encode(self, text: str) -> List[int]:
        """Token ids of one text WITHOUT [CLS] / [SEP]."""
        ids: List[int] = []
        for part in self._split_special(text):
            if isinstance(part, int):
                ids.append(part)
                continue
            for word in self._pre_tokenize(self._normalize(part)):
                ids.extend(self._wordpiece(word))
        return ids

    def __call__(self, texts: Union[str, Sequence[str]], max_length: Optional[int] = None, padding=False, truncation=False,
                 return_tensors: Optional[str] = None, **_) -> Encoding:
        single = isinstance(texts, str)
        if single:
            texts = [texts]
        rows = []
        for t in texts:
            ids = self.encode(t)
            if truncation and max_length is not None:
                ids = ids[: max(max_length - 2, 0)]
            rows.append([self.cls_token_id] + ids + [self.sep_token_id])
        if padding == "max_length" and max_length is not None:
            width = max_length
        elif padding in (True, "longest"):
            width = max((len(r) for r in rows), default=0)
        else:
            width = None
        if width is None and return_tensors is not None and len({len(r) for r in rows}) > 1:
            raise ValueError("rows of different lengths cannot be returned as a tensor: pass padding='max_length' or True")
        input_ids, mask = [], []
        for r in rows:
            pad = max((width or len(r)) - len(r), 0)
            input_ids.append(r + [self.pad_token_id] * pad)
            mask.append([1] * len(r) + [0] * pad)
        enc = {"input_ids": input_ids, "token_type_ids": [[0] * len(r) for r in input_ids], "attention_mask": mask}
        if return_tensors == "pt":
            enc = {k: torch.tensor(v, dtype=torch.int64).reshape(len(rows), -1) for k, v in enc.items()}
        elif return_tensors is not None:
            raise ValueError(f"return_tensors={return_tensors!r}: only 'pt' is supported")
        elif single:
            enc = {k: v[0] for k, v in enc.items()}      # a str in, flat lists out -- as the HF tokenizer does without return_tensors
        return Encoding(enc)


def load_tokenizer(model_path: str, backend: str = "native"):
    """The tokenizer of a release directory.  ``backend='native'`` (default): :class:`WordPieceTokenizer` over its ``vocab.txt``;
    ``backend='hf'``: ``AutoTokenizer.from_pretrained(model_path)`` restricted to local files (needs ``transformers``)."""
    if not os.path.isdir(model_path):
        raise FileNotFoundError(f"{model_path}: not a directory")
    has_vocab = os.path.exists(os.path.join(model_path, "vocab.txt"))
    if not has_vocab and not os.path.exists(os.path.join(model_path, "tokenizer.json")):
        raise FileNotFoundError(f"{model_path}: neither vocab.txt nor tokenizer.json found (the release directory ships "
                                "the PubMedBERT vocabulary; it is not bundled with keep_amd)")
    if backend == "native" and has_vocab:
        lower = True
        cfg = os.path.join(model_path, "tokenizer_config.json")
        if os.path.exists(cfg):
            import json
            with open(cfg) as f:
                lower = bool(json.load(f).get("do_lower_case", True))
        return WordPieceTokenizer(os.path.join(model_path, "vocab.txt"), do_lower_case=lower)
    if backend not in ("native", "hf"):
        raise ValueError(f"backend {backend!r}: 'native' or 'hf'")
    from transformers import AutoTokenizer, BertTokenizerFast
    if os.path.exists(os.path.join(model_path, "tokenizer_config.json")) or os.path.exists(os.path.join(model_path, "config.json")):
        try:
            return AutoTokenizer.from_pretrained(model_path, local_files_only=True, trust_remote_code=False)
        except Exception:       # a KEEP config.json names a custom model type AutoTokenizer cannot map: fall through
            pass
    return BertTokenizerFast.from_pretrained(model_path, local_files_only=True, do_lower_case=True)


def tokenize(tokenizer, texts: Sequence[str], max_length: int = MAX_LENGTH) -> Mapping:
    """The reference's tokenizer call: ``[CLS] ... [SEP] [PAD]*`` padded / truncated to ``max_length`` ->
    ``input_ids``, ``token_type_ids``, ``attention_mask`` as ``[P, max_length]`` int64 tensors."""
    return tokenizer(list(texts), max_length=max_length, padding="max_length", truncation=True, return_tensors="pt")

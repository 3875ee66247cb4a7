"""Tokenizer side of the text path.

The reference tokenises with the third-party HF tokenizer shipped in the release directory
(``quick_start/keep_inference.py:87``: ``AutoTokenizer.from_pretrained(model_path)``; PubMedBERT uncased WordPiece) and
always with the same call (``keep_inference.py:99``, ``WSI_evaluation/utils.py:73``).  Nothing here re-implements
WordPiece: ``load_tokenizer`` opens the same files with the same library (offline, without executing code from the
directory), ``tokenize`` is that one call, and the engine's contract starts at the three ``[P, T]`` int64 tensors it
returns."""
from __future__ import annotations

import os
from typing import Mapping, Sequence

MAX_LENGTH = 256       # keep_inference.py:99 / utils.py:73


def load_tokenizer(model_path: str):
    """``AutoTokenizer.from_pretrained(model_path)`` restricted to local files (``vocab.txt`` and/or ``tokenizer.json``
    in the release directory)."""
    if not os.path.isdir(model_path):
        raise FileNotFoundError(f"{model_path}: not a directory")
    if not any(os.path.exists(os.path.join(model_path, f)) for f in ("vocab.txt", "tokenizer.json")):
        raise FileNotFoundError(f"{model_path}: neither vocab.txt nor tokenizer.json found (the release directory ships "
                                "the PubMedBERT vocabulary; it is not bundled with keep_amd)")
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

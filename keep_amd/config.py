"""Shape description of the two KEEP towers.

Mirrors what the reference fixes in code rather than in a config file:
``quick_start/keep_inference.py:32-40`` (timm ``vit_large_patch16_224`` ctor
arguments), ``:42-46`` (visual_head), ``:49-50`` (``BertConfig(**text_config)``)
and ``:52`` (logit_scale).  SURVEY.md appendix A has the operator semantics.
"""
from __future__ import annotations

import dataclasses
import json
import math
import os
from typing import Any, Dict, Optional


@dataclasses.dataclass(frozen=True)
class VisionShape:
    """timm ``vit_large_patch16_224`` (UNI ViT-L/16) as built at keep_inference.py:32-40."""
    img_size: int = 224
    patch_size: int = 16
    embed_dim: int = 1024
    depth: int = 24
    num_heads: int = 16
    mlp_dim: int = 4096
    ln_eps: float = 1e-6

    @property
    def grid(self) -> int:
        return self.img_size // self.patch_size

    @property
    def num_patches(self) -> int:
        return self.grid * self.grid

    @property
    def num_tokens(self) -> int:          # CLS + patches
        return self.num_patches + 1

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.num_heads

    @property
    def patch_dim(self) -> int:           # 3 * 16 * 16
        return 3 * self.patch_size * self.patch_size


@dataclasses.dataclass(frozen=True)
class TextShape:
    """HF ``BertConfig`` fields the hot path reads (PubMedBERT-base shape)."""
    vocab_size: int = 30522
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    max_position_embeddings: int = 512
    type_vocab_size: int = 2
    layer_norm_eps: float = 1e-12

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads


@dataclasses.dataclass(frozen=True)
class KEEPShape:
    vision: VisionShape = VisionShape()
    text: TextShape = TextShape()
    projection_dim: int = 768             # keep_inference.py:15
    logit_scale_init: float = math.log(1 / 0.04)   # keep_inference.py:52

    @staticmethod
    def from_config_json(path_or_dict) -> "KEEPShape":
        """Read the release ``config.json`` (``KEEPConfig``, keep_inference.py:9-22).

        ``vision_config`` carries nothing the reference reads (the ViT ctor is
        hard-coded), so only ``text_config`` and ``projection_dim`` matter.
        """
        if isinstance(path_or_dict, (str, os.PathLike)):
            with open(path_or_dict) as f:
                d = json.load(f)
        else:
            d = dict(path_or_dict)
        tc: Dict[str, Any] = d.get("text_config") or {}
        t = TextShape(
            vocab_size=tc.get("vocab_size", 30522),
            hidden_size=tc.get("hidden_size", 768),
            num_hidden_layers=tc.get("num_hidden_layers", 12),
            num_attention_heads=tc.get("num_attention_heads", 12),
            intermediate_size=tc.get("intermediate_size", 3072),
            max_position_embeddings=tc.get("max_position_embeddings", 512),
            type_vocab_size=tc.get("type_vocab_size", 2),
            layer_norm_eps=tc.get("layer_norm_eps", 1e-12),
        )
        act = tc.get("hidden_act", "gelu")
        if act != "gelu":
            raise ValueError(f"text_config.hidden_act={act!r}: only erf-GELU is implemented")
        if tc.get("position_embedding_type", "absolute") != "absolute":
            raise ValueError("only absolute position embeddings are implemented")
        return KEEPShape(vision=VisionShape(), text=t,
                         projection_dim=d.get("projection_dim", 768))


def small_shape(vit_depth: int = 2, bert_layers: int = 2) -> KEEPShape:
    """Full-width, reduced-depth variant used by fast parity tests."""
    return KEEPShape(vision=dataclasses.replace(VisionShape(), depth=vit_depth),
                     text=dataclasses.replace(TextShape(), num_hidden_layers=bert_layers))


# Algorithmic work per unit, SURVEY.md §8(d) / BASELINE.md §2 (MAC = 2 FLOP).
def vit_flops_per_tile(v: VisionShape = VisionShape(), proj: int = 768) -> int:
    n, d, f = v.num_tokens, v.embed_dim, v.mlp_dim
    qkv = 2 * n * d * 3 * d
    att = 2 * 2 * v.num_heads * n * n * v.head_dim
    prj = 2 * n * d * d
    mlp = 2 * 2 * n * d * f
    pe = 2 * v.num_patches * v.patch_dim * d
    head = 2 * d * proj + 2 * proj * proj
    return v.depth * (qkv + att + prj + mlp) + pe + head


def bert_flops_per_prompt(t: TextShape = TextShape(), seq: int = 256) -> int:
    d, f = t.hidden_size, t.intermediate_size
    qkv = 2 * seq * d * 3 * d
    att = 2 * 2 * t.num_attention_heads * seq * seq * t.head_dim
    out = 2 * seq * d * d
    ffn = 2 * 2 * seq * d * f
    return t.num_hidden_layers * (qkv + att + out + ffn) + 2 * d * d


assert vit_flops_per_tile() == 123_110_129_664
assert bert_flops_per_prompt() == 45_903_642_624

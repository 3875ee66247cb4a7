"""Seeded synthetic KEEP weights in the release ``state_dict`` key layout.

There are no KEEP weights in this environment (no network), so every parity
test and the benchmark run on these.  The key layout is the one
``load_state_dict(strict=True)`` at ``quick_start/keep_inference.py:82-83``
expects.  Distributions follow SURVEY.md §8(d): linear
W~N(0,0.02..0.03), small biases, LN gamma 1±0.1, and LayerScale gamma drawn
from U(0.05,0.5) -- timm's init_values=1e-5 would make every block a no-op and
parity trivially true.
"""
from __future__ import annotations

from typing import Dict

import torch

from .config import KEEPShape


FAMILIES = ("default", "heavy_tail", "small_ls")


def synth_state_dict(shape: KEEPShape = KEEPShape(), seed: int = 0,
                     vision: bool = True, text: bool = True, family: str = "default") -> Dict[str, torch.Tensor]:
    """``family`` selects the weight distribution of the image tower (the text tower is the same in all three):
      default     W ~ N(0, 0.02-0.03), LN gamma 1 +- 0.1, LayerScale ~ U(0.05, 0.5)                      
      heavy_tail  trained-ViT pathologies: a few "massive" residual channels (cls / pos_embed / block biases 50-100 x the rest),
                  outlier LayerNorm gains (x4-8 on a few channels, x0.15 on the massive ones), heavy-tailed linear weights
                  (scale mixture: 4.5 % of the entries x3, 0.5 % x8, renormalised to the same std)
      small_ls    LayerScale ~ U(0.003, 0.03): blocks that barely move the residual stream (the regime just after timm's 1e-5 init)
    The default family draws exactly what it always drew (the committed fixtures depend on it); the others transform it with a second
    generator."""
    if family not in FAMILIES:
        raise ValueError(f"family {family!r}: one of {FAMILIES}")
    g = torch.Generator(device="cpu").manual_seed(seed)

    def normal(*size, std=0.02):
        return torch.randn(*size, generator=g, dtype=torch.float32) * std

    def uniform(*size, lo=0.0, hi=1.0):
        return torch.rand(*size, generator=g, dtype=torch.float32) * (hi - lo) + lo

    def ln(prefix, n, sd):
        sd[prefix + ".weight"] = 1.0 + normal(n, std=0.1)
        sd[prefix + ".bias"] = normal(n, std=0.05)

    def linear(prefix, out_f, in_f, sd, std=0.025):
        sd[prefix + ".weight"] = normal(out_f, in_f, std=std)
        sd[prefix + ".bias"] = normal(out_f, std=0.02)

    sd: Dict[str, torch.Tensor] = {}
    sd["logit_scale"] = torch.tensor(shape.logit_scale_init, dtype=torch.float32)
    if vision:
        v = shape.vision
        d = v.embed_dim
        sd["visual.cls_token"] = normal(1, 1, d)
        sd["visual.pos_embed"] = normal(1, v.num_tokens, d)
        sd["visual.patch_embed.proj.weight"] = normal(d, 3, v.patch_size, v.patch_size, std=0.03)
        sd["visual.patch_embed.proj.bias"] = normal(d, std=0.02)
        for i in range(v.depth):
            p = f"visual.blocks.{i}."
            ln(p + "norm1", d, sd)
            linear(p + "attn.qkv", 3 * d, d, sd, std=0.03)
            linear(p + "attn.proj", d, d, sd)
            sd[p + "ls1.gamma"] = uniform(d, lo=0.05, hi=0.5)
            ln(p + "norm2", d, sd)
            linear(p + "mlp.fc1", v.mlp_dim, d, sd)
            linear(p + "mlp.fc2", d, v.mlp_dim, sd, std=0.02)
            sd[p + "ls2.gamma"] = uniform(d, lo=0.05, hi=0.5)
        ln("visual.norm", d, sd)
        linear("visual_head.0", shape.projection_dim, d, sd)
        linear("visual_head.2", shape.projection_dim, shape.projection_dim, sd)
    if text:
        t = shape.text
        h = t.hidden_size
        sd["text.embeddings.word_embeddings.weight"] = normal(t.vocab_size, h)
        sd["text.embeddings.position_embeddings.weight"] = normal(t.max_position_embeddings, h)
        sd["text.embeddings.token_type_embeddings.weight"] = normal(t.type_vocab_size, h)
        ln("text.embeddings.LayerNorm", h, sd)
        for i in range(t.num_hidden_layers):
            p = f"text.encoder.layer.{i}."
            linear(p + "attention.self.query", h, h, sd, std=0.03)
            linear(p + "attention.self.key", h, h, sd, std=0.03)
            linear(p + "attention.self.value", h, h, sd)
            linear(p + "attention.output.dense", h, h, sd)
            ln(p + "attention.output.LayerNorm", h, sd)
            linear(p + "intermediate.dense", t.intermediate_size, h, sd)
            linear(p + "output.dense", h, t.intermediate_size, sd, std=0.02)
            ln(p + "output.LayerNorm", h, sd)
        linear("text.pooler.dense", h, h, sd)
    if vision and family != "default":
        _apply_family(sd, shape, family, seed)
    return sd


def _apply_family(sd: Dict[str, torch.Tensor], shape: KEEPShape, family: str, seed: int) -> None:
    g = torch.Generator(device="cpu").manual_seed(seed * 7919 + 17)
    v = shape.vision
    d = v.embed_dim
    if family == "small_ls":
        for i in range(v.depth):
            for k in ("ls1", "ls2"):
                sd[f"visual.blocks.{i}.{k}.gamma"] = torch.rand(d, generator=g) * 0.027 + 0.003
        return
    massive = torch.randperm(d, generator=g)[:6]
    sign = torch.where(torch.rand(6, generator=g) < 0.5, -1.0, 1.0)
    sd["visual.cls_token"][0, 0, massive] = sign * (1.0 + torch.rand(6, generator=g))               # 50-100 x the 0.02 of the others
    sd["visual.pos_embed"][0, :, massive] += (sign * (1.0 + torch.rand(6, generator=g)))[None, :]
    for i in range(v.depth):
        p = f"visual.blocks.{i}."
        for name in ("attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2"):
            w = sd[p + name + ".weight"]
            u = torch.rand(w.shape, generator=g)
            mix = torch.where(u < 0.005, 8.0, torch.where(u < 0.05, 3.0, 1.0))
            std0 = float(w.std())
            w.mul_(mix)
            w.mul_(std0 / float(w.std()))
        for name in ("attn.proj", "mlp.fc2"):                                                      # biases that keep feeding the massive channels
            sd[p + name + ".bias"][massive] = sign * (1.0 + 2.0 * torch.rand(6, generator=g))    # x LayerScale x 48 sub-blocks: those channels end 50-100 x the median |x|
        for name in ("norm1", "norm2"):
            gam = sd[p + name + ".weight"]
            out = torch.randperm(d, generator=g)[:8]
            gam[out] *= 4.0 + 4.0 * torch.rand(8, generator=g)
            gam[massive] = 0.15
    gam = sd["visual.norm.weight"]
    gam[torch.randperm(d, generator=g)[:8]] *= 4.0 + 4.0 * torch.rand(8, generator=g)


def synth_tiles(batch: int, seed: int = 0, dtype=torch.float32) -> torch.Tensor:
    """ImageNet-normalised tiles are ~N(0,1); SURVEY.md §8(d) config 2."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(batch, 3, 224, 224, generator=g, dtype=torch.float32).to(dtype)


def synth_prompts(n: int, seq: int = 256, seed: int = 1, vocab: int = 30522,
                  min_len: int = 8, max_len: int = 32) -> Dict[str, torch.Tensor]:
    """Token tensors shaped like the tokenizer call at keep_inference.py:99.

    ``[CLS] w.. [SEP] [PAD]*`` with valid lengths ~U{min_len..max_len}
    (SURVEY.md §8(d) config 3).  Ids are random: no vocab file exists here.
    """
    g = torch.Generator(device="cpu").manual_seed(seed)
    ids = torch.randint(4, vocab, (n, seq), generator=g, dtype=torch.int64)
    lens = torch.randint(min_len, min(max_len, seq) + 1, (n,), generator=g)
    pos = torch.arange(seq)[None, :]
    mask = (pos < lens[:, None]).to(torch.int64)
    ids[:, 0] = 2                                   # [CLS] in PubMedBERT's vocab
    ids[torch.arange(n), lens - 1] = 3              # [SEP]
    ids = ids * mask                                # [PAD] = 0
    return {"input_ids": ids, "token_type_ids": torch.zeros_like(ids), "attention_mask": mask}


def synth_tiles_device(a: int, b: int, device, dtype=torch.bfloat16, seed: int = 1000, unit: int = 256) -> torch.Tensor:
    """Tiles [a, b) of an endless synthetic slide, generated ON the device.

    The slide is defined in units of ``unit`` tiles (unit u = randn seeded with seed + u), so a tile's pixels depend
    only on its global index: every world size / shard boundary / batch size sees the same slide, and a batch costs
    one or two ``randn`` launches instead of a Python loop over tiles."""
    if b <= a:
        return torch.empty((0, 3, 224, 224), device=device, dtype=dtype)
    parts = []
    for u in range(a // unit, (b - 1) // unit + 1):
        g = torch.Generator(device=device).manual_seed(seed + u)
        block = torch.randn(unit, 3, 224, 224, device=device, generator=g, dtype=torch.float32)
        parts.append(block[max(a - u * unit, 0): min(b - u * unit, unit)].to(dtype))
    return parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)


def towers_of(state_dict) -> tuple:
    """Which towers a state_dict carries -- for ``KEEPModel(towers=...)`` when a single tower is loaded on purpose."""
    return tuple(t for t, pre in (("image", "visual."), ("text", "text.")) if any(k.startswith(pre) for k in state_dict))


SYNTH_WORDS = ("an h & e image of a histopathology slide showing tissue with features consistent . , chromophobe clear cell papillary renal "
               "carcinoma normal kidney tumor benign parenchyma kind type variant region").split()


def write_synthetic_release(directory: str, shape: KEEPShape = KEEPShape(), seed: int = 0, weights: str = "safetensors",
                            drop_keys=()) -> str:
    """A release directory in the on-disk format ``AutoModel.from_pretrained(model_path)`` / ``AutoTokenizer.from_pretrained(model_path)``
    open (zeroshot_subtyping_WSI.py:44-46, keep_inference.py:79-87): ``config.json`` (``model_type: keep``, ``text_config``,
    ``projection_dim``), ``model.safetensors`` or ``pytorch_model.bin`` with seeded weights in the key layout of SURVEY.md A.3, and an
    uncased WordPiece ``vocab.txt`` + ``tokenizer_config.json`` (a small word list padded to the text tower's vocabulary size).
    ``drop_keys``: state_dict keys left out on purpose (strict-loading tests).  Returns ``directory``."""
    import json
    import os
    os.makedirs(directory, exist_ok=True)
    sd = {k: v.contiguous() for k, v in synth_state_dict(shape, seed=seed).items() if k not in set(drop_keys)}
    if weights == "safetensors":
        from safetensors.torch import save_file
        save_file(sd, os.path.join(directory, "model.safetensors"))
    elif weights == "bin":
        torch.save(sd, os.path.join(directory, "pytorch_model.bin"))
    else:
        raise ValueError("weights: 'safetensors' or 'bin'")
    t = shape.text
    with open(os.path.join(directory, "config.json"), "w") as f:
        json.dump({"model_type": "keep", "projection_dim": shape.projection_dim, "vision_config": None,
                   "text_config": {"vocab_size": t.vocab_size, "hidden_size": t.hidden_size, "num_hidden_layers": t.num_hidden_layers,
                                   "num_attention_heads": t.num_attention_heads, "intermediate_size": t.intermediate_size,
                                   "max_position_embeddings": t.max_position_embeddings, "type_vocab_size": t.type_vocab_size,
                                   "layer_norm_eps": t.layer_norm_eps, "hidden_act": "gelu"}}, f)
    alnum = "abcdefghijklmnopqrstuvwxyz0123456789"
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + sorted(set(SYNTH_WORDS)) + [f"##{c}" for c in alnum] + list(alnum)
    vocab += [f"[unused{i}]" for i in range(t.vocab_size - len(vocab))]
    with open(os.path.join(directory, "vocab.txt"), "w") as f:
        f.write("\n".join(vocab) + "\n")
    with open(os.path.join(directory, "tokenizer_config.json"), "w") as f:
        json.dump({"tokenizer_class": "BertTokenizer", "do_lower_case": True, "model_max_length": t.max_position_embeddings}, f)
    return directory


def synthetic_rcc_prompts(prompt_sets: int = 48, seed: int = 3) -> Dict[str, dict]:
    """A prompt file with the structure the WSI scripts read (``prompts[str(i)] = {'classnames': {label: str}, 'templates': str}``,
    zeroshot_subtyping_WSI.py:31-36, utils.py:86-104), drawn from the words of the synthetic vocabulary."""
    import random
    names = {"CHRCC": ["chromophobe renal cell carcinoma", "renal carcinoma chromophobe type"],
             "CCRCC": ["clear cell renal cell carcinoma", "renal carcinoma clear cell type"],
             "PRCC": ["papillary renal cell carcinoma", "renal carcinoma papillary variant"],
             "Normal": ["normal kidney tissue", "benign renal parenchyma"]}
    templates = ["an H&E image of CLASSNAME.", "a histopathology slide showing CLASSNAME.", "tissue with features consistent with CLASSNAME."]
    rnd = random.Random(seed)
    return {str(i): {"classnames": {k: rnd.choice(v) for k, v in names.items()}, "templates": rnd.choice(templates)} for i in range(prompt_sets)}

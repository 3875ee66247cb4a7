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

import math
import os
from typing import Dict, Optional

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


# ------------------------------------------------------------------------------------------------------------------
# Structured tile families: pixels that look like what the reference transform (keep_inference.py:88-93; the WSI scripts' tiles,
# zeroshot_subtyping_WSI.py:47-52) hands to encode_image -- uint8 RGB tiles, spatially correlated, non-zero channel means after the
# ImageNet normalisation, and slides full of near-constant background -- instead of the i.i.d. N(0,1) pixels of BASELINE config 2.
# Every family is generated ON the device as uint8 [n,224,224,3] (HWC, what encode_image_uint8 takes); a tile's pixels depend only on
# (family, seed, global tile index // unit), like synth_tiles_device.
# ------------------------------------------------------------------------------------------------------------------
TILE_FAMILIES = ("he_crops", "stain_field", "background", "half", "mixed")
_STAIN_H = (0.65, 0.70, 0.29)      # optical-density vectors of haematoxylin / eosin (Ruifrok & Johnston colour deconvolution)
_STAIN_E = (0.07, 0.99, 0.11)


def _octave_field(n: int, g: torch.Generator, device, beta: float = 1.0, octaves: int = 7, size: int = 224) -> torch.Tensor:
    """[n,1,size,size] low-pass ("1/f^beta") noise, zero mean / unit variance per tile: a sum of bilinearly up-sampled white-noise grids,
    octave o (a 2^(o+1)+1 point grid) weighted 2^(-o beta)."""
    import torch.nn.functional as F
    out = torch.zeros(n, 1, size, size, device=device)
    for o in range(octaves):
        k = 2 ** (o + 1) + 1
        grid = torch.randn(n, 1, k, k, device=device, generator=g)
        out += F.interpolate(grid, size=(size, size), mode="bilinear", align_corners=True) * (2.0 ** (-o * beta))
    out -= out.mean(dim=(2, 3), keepdim=True)
    return out / out.std(dim=(2, 3), keepdim=True).clamp_min(1e-6)


def _stain_field_tiles(n: int, g: torch.Generator, device) -> torch.Tensor:
    """H & E by Beer-Lambert: RGB = 255 * 10^-(c_H v_H + c_E v_E) with two correlated low-pass concentration fields (nuclei = peaks of a
    sharper field), per-tile staining strength, sensor noise.  float [n,3,224,224] in 0..255 (not yet rounded)."""
    base, fine = _octave_field(n, g, device, beta=1.2), _octave_field(n, g, device, beta=0.4)
    strength = 0.6 + 0.8 * torch.rand(n, 1, 1, 1, device=device, generator=g)
    c_e = (0.35 + 0.25 * base).clamp_min(0.0) * strength                                  # eosin: stroma / cytoplasm, smooth
    c_h = (0.9 * torch.sigmoid(4.0 * (0.6 * fine + 0.4 * base) - 3.0)) * strength         # haematoxylin: sparse dark nuclei
    vh = torch.tensor(_STAIN_H, device=device).view(1, 3, 1, 1)
    ve = torch.tensor(_STAIN_E, device=device).view(1, 3, 1, 1)
    rgb = 255.0 * torch.pow(10.0, -(c_h * vh * 1.4 + c_e * ve * 0.6))
    return rgb + 2.0 * torch.randn(n, 3, 224, 224, device=device, generator=g)


def _background_tiles(n: int, g: torch.Generator, device) -> torch.Tensor:
    """What most of a slide is: glass.  45 % near-white with sensor noise, 25 % exactly 255 (saturated), 15 % light grey with a tint,
    10 % black (outside the scanned area / pen), 5 % a flat mid-grey.  float [n,3,224,224] in 0..255."""
    kind = torch.rand(n, device=device, generator=g)
    level = torch.where(kind < 0.45, 244.0 + 8.0 * torch.rand(n, device=device, generator=g),
                        torch.where(kind < 0.70, torch.full((n,), 255.0, device=device),
                                    torch.where(kind < 0.85, 215.0 + 25.0 * torch.rand(n, device=device, generator=g),
                                                torch.where(kind < 0.95, torch.zeros(n, device=device), torch.full((n,), 128.0, device=device)))))
    tint = torch.where(((kind >= 0.70) & (kind < 0.85))[:, None], 6.0 * (torch.rand(n, 3, device=device, generator=g) - 0.5), torch.zeros(n, 3, device=device))
    sigma = torch.where((kind >= 0.45) & (kind < 0.70), 0.0, 1.5)
    noise = torch.randn(n, 3, 224, 224, device=device, generator=g) * sigma.view(n, 1, 1, 1)
    return (level.view(n, 1, 1, 1) + tint.view(n, 3, 1, 1)).expand(n, 3, 224, 224) + noise


def _he_crop_tiles(n: int, g: torch.Generator, device, base_image: torch.Tensor) -> torch.Tensor:
    """Random resized crops (scale 0.45-1 of the short side), flips, quarter turns and a mild colour jitter of ONE real H & E image
    (``base_image``: uint8 [H,W,3], e.g. the reference's quick_start/example.tif).  float [n,3,224,224] in 0..255."""
    import torch.nn.functional as F
    img = base_image.to(device=device, dtype=torch.float32).permute(2, 0, 1)[None]
    H, W = img.shape[2], img.shape[3]
    u = torch.rand(n, 8, device=device, generator=g)
    side = (0.45 + 0.55 * u[:, 0]) * min(H, W)                                           # crop side in pixels
    cx = side / 2 + u[:, 1] * (W - side)
    cy = side / 2 + u[:, 2] * (H - side)
    quarter = torch.floor(u[:, 3] * 4.0) * (math.pi / 2)
    flip = torch.where(u[:, 4] < 0.5, -1.0, 1.0)
    cos, sin = torch.cos(quarter), torch.sin(quarter)
    sx, sy = side / W, side / H                                                           # normalised half-extents
    theta = torch.stack([torch.stack([sx * cos * flip, -sx * sin, 2 * cx / W - 1], 1),
                         torch.stack([sy * sin * flip, sy * cos, 2 * cy / H - 1], 1)], 1)
    grid = F.affine_grid(theta, (n, 3, 224, 224), align_corners=False)
    crops = F.grid_sample(img.expand(n, -1, -1, -1), grid, mode="bilinear", padding_mode="border", align_corners=False)
    gain = 0.9 + 0.2 * torch.rand(n, 3, 1, 1, device=device, generator=g)
    offset = 16.0 * (torch.rand(n, 3, 1, 1, device=device, generator=g) - 0.5)
    return (crops - 128.0) * gain + 128.0 + offset


def _default_base_image() -> torch.Tensor:
    """The one real tile this repository holds: the reference's quick-start image, kept as a data fixture (tests/golden/example.tif)."""
    import numpy as np
    from PIL import Image
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "example.tif")
    return torch.from_numpy(np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8).copy())


def synth_tile_family(family: str, a: int, b: int, device, seed: int = 7000, unit: int = 256,
                      base_image: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Tiles [a, b) of an endless synthetic slide of one of ``TILE_FAMILIES``, uint8 [b-a,224,224,3] on ``device``:

      he_crops     random resized crops / flips / quarter turns / colour jitter of a real H & E image (``base_image``, default example.tif)
      stain_field  Beer-Lambert H & E from two low-pass (1/f) stain-concentration fields: correlated pixels, pink-purple channel means
      background   glass: near-white + noise, saturated 255, tinted light grey, black, flat grey -- near-constant tiles
      half         one side of a random straight edge is background, the other stain_field
      mixed        tile i is of family (he_crops, stain_field, background, half)[i % 4]: a slide

    ``normalise_u8`` turns them into the float tiles ``encode_image`` takes (ToTensor + Normalize, keep_inference.py:91-92)."""
    if family not in TILE_FAMILIES:
        raise ValueError(f"family {family!r}: one of {TILE_FAMILIES}")
    if b <= a:
        return torch.empty((0, 224, 224, 3), device=device, dtype=torch.uint8)
    if family == "he_crops" or family == "mixed":
        base_image = _default_base_image() if base_image is None else base_image
    fid = TILE_FAMILIES.index(family)
    parts = []
    for u in range(a // unit, (b - 1) // unit + 1):
        g = torch.Generator(device=device).manual_seed(seed + 1_000_003 * fid + u)
        if family == "he_crops":
            x = _he_crop_tiles(unit, g, device, base_image)
        elif family == "stain_field":
            x = _stain_field_tiles(unit, g, device)
        elif family == "background":
            x = _background_tiles(unit, g, device)
        elif family == "half":
            tex, bg = _stain_field_tiles(unit, g, device), _background_tiles(unit, g, device)
            ang = 2 * math.pi * torch.rand(unit, 1, 1, device=device, generator=g)
            off = 60.0 * (torch.rand(unit, 1, 1, device=device, generator=g) - 0.5)
            yy, xx = torch.meshgrid(torch.arange(224.0, device=device) - 111.5, torch.arange(224.0, device=device) - 111.5, indexing="ij")
            side = (xx[None] * torch.cos(ang) + yy[None] * torch.sin(ang) + off) > 0
            x = torch.where(side[:, None], tex, bg)
        else:
            quarter = unit // 4
            x = torch.cat([_he_crop_tiles(quarter, g, device, base_image), _stain_field_tiles(quarter, g, device),
                           _background_tiles(quarter, g, device), _stain_field_tiles(unit - 3 * quarter, g, device)], 0)
            bgm = _background_tiles(unit - 3 * quarter, g, device)
            edge = (torch.arange(224.0, device=device) - 111.5)[None, None, None, :] > 40.0 * (torch.rand(unit - 3 * quarter, 1, 1, 1, device=device, generator=g) - 0.5)
            x[3 * quarter:] = torch.where(edge, x[3 * quarter:], bgm)
            # interleave: tile i of the unit is of family i % 4
            order = torch.arange(unit, device=device)
            src = (order % 4) * quarter + order // 4
            x = x[src.clamp_max(unit - 1)]
        x = x.clamp_(0.0, 255.0).round_().to(torch.uint8).permute(0, 2, 3, 1).contiguous()
        parts.append(x[max(a - u * unit, 0): min(b - u * unit, unit)])
    return parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)


def normalise_u8(tiles_u8: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
    """ToTensor + Normalize(ImageNet mean / std) of the reference transform (keep_inference.py:91-92) on uint8 [n,224,224,3] -> [n,3,224,224]."""
    from .preprocess import IMAGENET_MEAN, IMAGENET_STD
    x = tiles_u8.permute(0, 3, 1, 2).to(torch.float32) / 255.0
    mean = torch.tensor(IMAGENET_MEAN, dtype=torch.float32, device=x.device).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, dtype=torch.float32, device=x.device).view(1, 3, 1, 1)
    return ((x - mean) / std).to(dtype)


PROBE_FAMILIES = ("gaussian", "stain_field", "background", "half")


def calibration_probe(n_per_family: int, device, seed: int = 20250929, families=PROBE_FAMILIES, dtype=torch.bfloat16):
    """The probe ``KEEPModel.calibrate`` / ``calibrate_bias`` use when the caller passes no tiles of their own: ``n_per_family`` tiles of each of
    ``families`` -- i.i.d. N(0,1) pixels (BASELINE config 2) AND the structured families above (none needs a file; ``he_crops``, the real-image
    one, stays out on purpose: it is the held-out family the tests evaluate on).  -> (tiles [n,3,224,224] ``dtype``, normalised as the reference
    transform does; group index of every tile, int64 [n] on the CPU; the family names)."""
    parts, groups = [], []
    for gi, fam in enumerate(families):
        if fam == "gaussian":
            g = torch.Generator(device=device).manual_seed(seed)
            x = torch.randn(n_per_family, 3, 224, 224, device=device, generator=g, dtype=torch.float32).to(dtype)
        else:
            x = normalise_u8(synth_tile_family(fam, 0, n_per_family, device, seed=seed % 1_000_003), dtype)
        parts.append(x)
        groups.append(torch.full((n_per_family,), gi, dtype=torch.int64))
    return torch.cat(parts, 0), torch.cat(groups, 0), tuple(families)

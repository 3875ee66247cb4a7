#!/usr/bin/env python3
"""Pin the image-tower oracle to timm ITSELF the first time an environment has it.

The reference builds its vision encoder with ``timm.create_model("vit_large_patch16_224", pretrained=False, img_size=224,
patch_size=16, init_values=1e-5, num_classes=0, dynamic_img_size=True)`` (quick_start/keep_inference.py:32-40) and a
``visual_head`` of ``Linear(1024, 768) -> GELU -> Linear(768, 768)`` (:42-46); ``encode_image`` is
``normalize(visual_head(visual(x)), dim=-1)`` (:54-58).  timm (pinned 1.0.15, training/requirements.txt:13) is not installed in the
build container, so ``tests/golden/vit_d*.npz`` carry two stand-in pins (``features`` = transformers' Dinov2Model as ViT-L/16,
``features_aten_timm`` = timm's module tree restated on ATen ops).  This script adds the real one:

    pip install timm==1.0.15        # wherever that is possible
    python tools/pin_against_timm.py

For each ViT fixture it regenerates the seeded weights and tiles (checksums must match the fixture), builds the timm model exactly as the
reference does (depth overridden for the two-block fixture only), loads the seeded ``visual.*`` tensors with ``strict=True`` (which pins the
release key layout against timm's parameter names), runs the fp32 CPU forward, asserts that ``oracle/keep_oracle.py`` agrees to 1e-6
and writes ``features_timm`` (and ``timm_version``) into the fixture.  ``tests/test_oracle_golden.py`` and
``tests/test_towers_gpu.py::test_bench_weights_vs_both_image_tower_pins`` check that key whenever it is present.

Without timm the script says so and exits with status 3 (nothing is written).  It reads nothing under /root/reference.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from keep_amd.config import KEEPShape, small_shape                      # noqa: E402
from keep_amd.synth import synth_state_dict, synth_tiles                 # noqa: E402
from oracle import keep_oracle as O                                      # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
FIXTURES = ("vit_d2.npz", "vit_d24.npz", "vit_d24_bench.npz")
TOL = 1e-6


def checksum(t: torch.Tensor) -> float:
    return float(t.double().abs().sum())


def timm_keep_image_side(timm, sd, depth: int):
    """The reference's vision encoder + head on the seeded weights (quick_start/keep_inference.py:32-46)."""
    kw = dict(pretrained=False, img_size=224, patch_size=16, init_values=1e-5, num_classes=0, dynamic_img_size=True)
    if depth != 24:
        kw["depth"] = depth              # the two-block fixture: same model class, fewer blocks
    visual = timm.create_model("vit_large_patch16_224", **kw).eval()
    visual.load_state_dict({k[len("visual."):]: v for k, v in sd.items() if k.startswith("visual.")}, strict=True)
    head = torch.nn.Sequential(torch.nn.Linear(visual.num_features, 768), torch.nn.GELU(), torch.nn.Linear(768, 768)).eval()
    head.load_state_dict({k[len("visual_head."):]: v for k, v in sd.items() if k.startswith("visual_head.")}, strict=True)
    return visual, head


def pin(timm, name: str) -> float:
    path = os.path.join(GOLD, name)
    g = dict(np.load(path))
    depth = int(g["depth"])
    shape = KEEPShape() if depth == 24 else small_shape(vit_depth=depth)
    sd = synth_state_dict(shape, seed=int(g["weight_seed"]), text=False)
    x = synth_tiles(int(g["batch"]), seed=int(g["tile_seed"]))
    assert abs(checksum(x) - float(g["tiles_checksum"])) <= 1e-12 * float(g["tiles_checksum"]), "seeded tile generator drifted"
    assert abs(checksum(sd["visual.blocks.0.attn.qkv.weight"]) - float(g["qkv0_checksum"])) <= 1e-12 * float(g["qkv0_checksum"])
    visual, head = timm_keep_image_side(timm, sd, depth)
    with torch.no_grad():
        feat_timm = torch.nn.functional.normalize(head(visual(x)), dim=-1)          # keep_inference.py:54-58
        feat_or = O.encode_image(sd, x)
    d = float((feat_or - feat_timm).abs().max())
    d_hf = float(np.abs(feat_timm.numpy() - g["features"]).max())
    d_at = float(np.abs(feat_timm.numpy() - g["features_aten_timm"]).max())
    print(f"[{name}] oracle vs timm {timm.__version__}: max|dfeat| = {d:.3e}; timm vs the Dinov2 pin {d_hf:.3e}, vs the ATen-op restatement {d_at:.3e}")
    if not d <= TOL:
        raise SystemExit(f"{name}: oracle/keep_oracle.py disagrees with timm by {d:.3e} (> {TOL:g}): the oracle does NOT restate the reference's image tower")
    g["features_timm"] = feat_timm.numpy()
    g["oracle_dfeat_timm"] = np.float64(d)
    g["timm_version"] = np.array(timm.__version__)
    np.savez_compressed(path, **g)
    return d


def main() -> int:
    try:
        import timm
    except ImportError:
        print("timm is not installed here: the image tower stays pinned to its two stand-ins (DESIGN.md section 2). "
              "Run this script wherever `pip install timm==1.0.15` is possible.", file=sys.stderr)
        return 3
    worst = max(pin(timm, name) for name in FIXTURES)
    print(f"image tower pinned to timm {timm.__version__}: worst max|dfeat| {worst:.3e}; commit tests/golden/vit_d*.npz")
    return 0


if __name__ == "__main__":
    sys.exit(main())

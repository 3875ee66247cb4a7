"""Host preprocessing (reference transform, keep_inference.py:88-93) on the reference's own example tile."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from keep_amd.preprocess import IMAGENET_MEAN, IMAGENET_STD, preprocess, preprocess_batch


def test_example_tif_is_crop_scale_normalise(golden_dir):
    path = os.path.join(golden_dir, "example.tif")
    img = Image.open(path).convert("RGB")
    assert img.size == (298, 224)                      # resize(224) is the identity, crop takes columns 37..260
    x = preprocess(path)
    assert x.shape == (3, 224, 224) and x.dtype == torch.float32
    raw = np.asarray(img, dtype=np.float32)[:, 37:261, :] / 255.0
    ref = (torch.from_numpy(raw).permute(2, 0, 1) - torch.tensor(IMAGENET_MEAN)[:, None, None]) / torch.tensor(IMAGENET_STD)[:, None, None]
    assert torch.equal(x, ref)
    assert preprocess_batch([path, img]).shape == (2, 3, 224, 224)


def test_resize_then_crop_shapes():
    for w, h in ((512, 300), (300, 512), (224, 224), (100, 180)):
        arr = (np.random.default_rng(0).random((h, w, 3)) * 255).astype(np.uint8)
        assert preprocess(Image.fromarray(arr)).shape == (3, 224, 224)


def test_resample_restatement_is_pil_bit_for_bit():
    """keep_amd.preprocess restates Pillow's 8-bit bicubic resample (float64 window construction, 22-bit weights, two integer
    passes): up- and down-scaling, both orientations, the identity case."""
    from keep_amd.preprocess import resize_bicubic_u8_numpy, resize_output_size
    rng = np.random.default_rng(1)
    for w, h in ((512, 300), (300, 512), (1024, 1024), (256, 256), (100, 180), (231, 224), (897, 673)):
        arr = (rng.random((h, w, 3)) * 255).astype(np.uint8)
        ow, oh = resize_output_size(w, h)
        ref = np.asarray(Image.fromarray(arr).resize((ow, oh), Image.BICUBIC))
        assert np.array_equal(resize_bicubic_u8_numpy(arr, ow, oh), ref), (w, h)


@pytest.mark.gpu
def test_resize_crop_on_device_is_pil_bit_for_bit(golden_dir):
    """Row f4: Resize(224, bicubic) + CenterCrop on the device for raw uint8 tiles == the PIL path of keep_amd.preprocess
    (uint8 equality), and encode_image_raw == encode_image(preprocess(...)) to fp32 rounding."""
    from keep_amd import KEEPModel
    from keep_amd.config import small_shape
    from keep_amd.preprocess import _center_crop, _resize_shorter_side
    from keep_amd.synth import synth_state_dict
    sd = synth_state_dict(small_shape(2, 1), seed=8, text=False)
    m = KEEPModel(precision="strict", towers=("image",))
    m.load_state_dict(sd)
    m.to("cuda:0")
    rng = np.random.default_rng(2)
    for w, h, b in ((512, 300, 3), (300, 512, 2), (1024, 1024, 2), (256, 256, 5), (225, 224, 1), (897, 673, 1)):
        arr = (rng.random((b, h, w, 3)) * 255).astype(np.uint8)
        ref = np.stack([np.asarray(_center_crop(_resize_shorter_side(Image.fromarray(a), 224), 224)) for a in arr])
        got = m.resize_crop_uint8(torch.from_numpy(arr).cuda())
        assert got.shape == (b, 224, 224, 3) and np.array_equal(got.cpu().numpy(), ref), (w, h)
    imgs = [Image.fromarray(a) for a in arr]
    feats = m.encode_image_raw(torch.from_numpy(arr))
    ref_feats = m.encode_image(preprocess_batch(imgs))
    assert (feats - ref_feats).abs().max() < 2e-6
    tif = np.asarray(Image.open(os.path.join(golden_dir, "example.tif")).convert("RGB"))[None]
    assert (m.encode_image_raw(torch.from_numpy(tif.copy())) - m.encode_image(preprocess(os.path.join(golden_dir, "example.tif"))[None])).abs().max() < 2e-6
    with pytest.raises(ValueError):
        m.resize_crop_uint8(torch.zeros(1, 10, 10, 4, dtype=torch.uint8))


@pytest.mark.gpu
def test_quick_start_plumbing_matches_oracle(golden_dir):
    """BASELINE config 1: one tile x three prompts through both towers and the similarity."""
    from keep_amd import KEEPModel
    from keep_amd.config import small_shape
    from keep_amd.synth import synth_prompts, synth_state_dict
    from oracle import keep_oracle as O
    sd = synth_state_dict(small_shape(2, 2), seed=3)
    m = KEEPModel(precision="strict")
    m.load_state_dict(sd)
    m.to("cuda:0").eval()
    img = preprocess(os.path.join(golden_dir, "example.tif")).unsqueeze(0)
    toks = synth_prompts(3, 256, seed=4)
    sim = m.encode_image(img) @ m.encode_text(toks).T
    with torch.no_grad():
        ref = O.similarity(O.encode_image(sd, img), O.encode_text(sd, toks))
    assert sim.shape == (1, 3) and (sim - ref).abs().max() < 5e-6


@pytest.mark.gpu
def test_uint8_tiles_normalised_on_device(golden_dir):
    """raw uint8 HWC tiles, ToTensor + Normalize fused into the first kernel."""
    from keep_amd import KEEPModel
    from keep_amd.config import small_shape
    from keep_amd.synth import synth_state_dict
    sd = synth_state_dict(small_shape(2, 1), seed=8, text=False)
    m = KEEPModel(precision="strict", towers=("image",))
    m.load_state_dict(sd)
    m.to("cuda:0")
    img = Image.open(os.path.join(golden_dir, "example.tif")).convert("RGB").crop((37, 0, 261, 224))
    g = torch.Generator().manual_seed(1)
    u8 = torch.cat([torch.from_numpy(np.asarray(img, dtype=np.uint8).copy())[None],
                    torch.randint(0, 256, (4, 224, 224, 3), generator=g, dtype=torch.uint8)])
    mean, std = torch.tensor(IMAGENET_MEAN)[None, :, None, None], torch.tensor(IMAGENET_STD)[None, :, None, None]
    f32 = (u8.permute(0, 3, 1, 2).float() / 255.0 - mean) / std
    assert torch.equal(f32[0], preprocess(os.path.join(golden_dir, "example.tif")))
    a = m.encode_image_uint8(u8)
    b = m.encode_image(f32)
    assert (a - b).abs().max() < 2e-6
    with pytest.raises(ValueError):
        m.encode_image_uint8(f32)

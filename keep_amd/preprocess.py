"""Host-side tile preprocessing: the ``transforms.Compose`` of the reference
(``quick_start/keep_inference.py:88-93``, repeated in every WSI script and ``README.md:52-57``):

    Resize(size=224, interpolation=BICUBIC) -> CenterCrop((224, 224)) -> ToTensor() -> Normalize(ImageNet mean/std)

torchvision is not installed in this image, so this restates its PIL code path with PIL + numpy (torchvision
itself calls ``PIL.Image.resize(..., BICUBIC)`` for PIL inputs).  Parity with torchvision is unpinned for a
real resize; for the reference's ``example.tif`` (298x224) the resize is the identity and only crop + scaling remain.
"""
from __future__ import annotations

from typing import Iterable, Union

import numpy as np
import torch
from PIL import Image

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def _resize_shorter_side(img: Image.Image, size: int) -> Image.Image:
    w, h = img.size
    if (w <= h and w == size) or (h <= w and h == size):
        return img
    if w < h:
        ow, oh = size, int(size * h / w)
    else:
        oh, ow = size, int(size * w / h)
    return img.resize((ow, oh), Image.BICUBIC)


def _center_crop(img: Image.Image, size: int) -> Image.Image:
    w, h = img.size
    if w < size or h < size:                    # torchvision pads with zeros first
        padded = Image.new(img.mode, (max(w, size), max(h, size)))
        padded.paste(img, ((max(w, size) - w) // 2, (max(h, size) - h) // 2))
        img, (w, h) = padded, padded.size
    top = int(round((h - size) / 2.0))
    left = int(round((w - size) / 2.0))
    return img.crop((left, top, left + size, top + size))


def preprocess(img: Union[str, Image.Image], size: int = 224) -> torch.Tensor:
    """PIL image (or path) -> float32 [3, size, size], as the reference's ``transform(Image.open(p).convert('RGB'))``."""
    if not isinstance(img, Image.Image):
        img = Image.open(img)
    img = _center_crop(_resize_shorter_side(img.convert("RGB"), size), size)
    x = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).to(torch.float32) / 255.0
    mean = torch.tensor(IMAGENET_MEAN, dtype=torch.float32)[:, None, None]
    std = torch.tensor(IMAGENET_STD, dtype=torch.float32)[:, None, None]
    return (x - mean) / std


def preprocess_batch(images: Iterable[Union[str, Image.Image]], size: int = 224) -> torch.Tensor:
    return torch.stack([preprocess(i, size) for i in images])


# ------------------------------------------------------------------------------------------------------------------
# Pillow's 8-bit bicubic resample, restated (what ``Image.resize(..., BICUBIC)`` -- and therefore torchvision's
# ``Resize`` on PIL inputs, keep_inference.py:89 -- computes): per output coordinate a window of input pixels
# [xmin, xmin + n) with weights bicubic((x - center + 0.5) / filterscale) normalised to 1 and rounded to 22-bit fixed point
# (the support widens with the scale factor when shrinking = antialiasing); horizontal pass, uint8 rounding, vertical pass.
# The coefficient tables are built here in float64 exactly as libImaging does; the two integer passes run on the device
# (keep_resize_crop_u8), so the on-device result is bit-identical to PIL's (tests/test_preprocess.py).
# ------------------------------------------------------------------------------------------------------------------
PIL_PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_bicubic_coeffs(in_size: int, out_size: int):
    """-> (bounds int32 [out,2] = (first input index, count), weights int32 [out,ksize], ksize)."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.array([_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)], dtype=np.float64)
        ww = w.sum()
        if ww != 0.0:
            w = w / ww
        fixed = w * (1 << PIL_PRECISION_BITS)
        kk[xx, :xmax] = np.where(fixed < 0, (fixed - 0.5).astype(np.int64), (fixed + 0.5).astype(np.int64))   # C cast: toward zero
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def resize_output_size(w: int, h: int, size: int = 224):
    """torchvision ``Resize(size)``: shorter side -> size, the other int(size * long / short); identity when already there."""
    if (w <= h and w == size) or (h <= w and h == size):
        return w, h
    return (size, int(size * h / w)) if w < h else (int(size * w / h), size)


def resize_bicubic_u8_numpy(arr: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """The two integer passes on the host (test restatement of the device kernels): uint8 [H,W,C] -> uint8 [out_h,out_w,C]."""
    h, w, _ = arr.shape
    src = arr.astype(np.int64)
    if out_w != w:
        bounds, kk, _ = pil_bicubic_coeffs(w, out_w)
        tmp = np.empty((h, out_w, arr.shape[2]), dtype=np.int64)
        for xx in range(out_w):
            x0, n = bounds[xx]
            acc = (src[:, x0:x0 + n, :] * kk[xx, :n, None].astype(np.int64)).sum(axis=1) + (1 << (PIL_PRECISION_BITS - 1))
            tmp[:, xx, :] = np.clip(acc >> PIL_PRECISION_BITS, 0, 255)
        src = tmp
    if out_h != h:
        bounds, kk, _ = pil_bicubic_coeffs(h, out_h)
        out = np.empty((out_h, src.shape[1], arr.shape[2]), dtype=np.int64)
        for yy in range(out_h):
            y0, n = bounds[yy]
            acc = (src[y0:y0 + n, :, :] * kk[yy, :n, None, None].astype(np.int64)).sum(axis=0) + (1 << (PIL_PRECISION_BITS - 1))
            out[yy] = np.clip(acc >> PIL_PRECISION_BITS, 0, 255)
        src = out
    return src.astype(np.uint8)

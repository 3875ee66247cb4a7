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

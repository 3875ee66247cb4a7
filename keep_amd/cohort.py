"""Cohort mode: the reference's feature-file dataset and ``run(classifier, dataloader, device)`` loops
 on the GPU similarity kernels.

  * ``WSIClassificationDataset``  <- ``WSI_Classification_Dataset`` (WSI_evaluation/utils.py:11-61):
    one slide per item, CLAM-style feature files ``<data_source>/pt_files/<slide>.pt`` (tensor [N,768])
    or ``<data_source>/h5_files/<slide>.h5`` (datasets ``features`` [N,768] f32, ``coords`` [N,2]).
    h5py is imported lazily: it is not installed in this image, the ``.pt`` path needs nothing.
  * ``run_subtyping`` / ``run_detection`` / ``run_segmentation`` <- the three ``run(classifier, dataloader, device)``
    functions, same arguments and return values (subtyping_utils.py:12-35 raw cosine logits; detection_utils.py:12-36 and
    segment_utils.py:16-42 softmax(10*logits)); ``keep_amd/wsi_evaluation/*_utils.py`` export each of them as ``run``.
  * ``save_slide_features`` writes what ``encode_image`` produced in the same on-disk formats, so feature
    files can be regenerated with this engine instead of the offline CLAM extraction (README.md:74).
"""
from __future__ import annotations

import os
from typing import Dict, Mapping, Optional, Sequence

import numpy as np
import torch

from .wsi import _engine, _normalized


class WSIClassificationDataset(torch.utils.data.Dataset):
    def __init__(self, df, data_source, target_transform=None, index_col="slide_id", target_col="Diagnosis",
                 use_h5=True, label_map=None):
        self.label_map, self.data_source = label_map, data_source
        self.index_col, self.target_col, self.target_transform = index_col, target_col, target_transform
        self.data, self.use_h5 = df, use_h5

    def __len__(self):
        return len(self.data)

    def _cell(self, idx, col):
        d = self.data
        return d.loc[idx, col] if hasattr(d, "loc") else d[idx][col]

    def get_ids(self, ids):
        return str(self._cell(ids, self.index_col))

    def get_labels(self, ids):
        return self._cell(ids, self.target_col)

    def __getitem__(self, idx):
        slide_id = str(self.get_ids(idx))
        label = self.get_labels(idx)
        if self.label_map is not None:
            label = self.label_map[label]
        if self.target_transform is not None:
            label = self.target_transform(label)
        if self.use_h5:
            try:
                import h5py
            except ImportError as e:            # pragma: no cover - h5py absent in the build image
                raise ImportError("use_h5=True needs h5py; use the pt_files layout (use_h5=False) instead") from e
            with h5py.File(os.path.join(self.data_source, "h5_files", slide_id + ".h5"), "r") as f:
                features = torch.from_numpy(f["features"][:])
                coords = torch.from_numpy(f["coords"][:])
        else:
            features = torch.load(os.path.join(self.data_source, "pt_files", slide_id + ".pt"))
            coords = []
        return {"features": features, "coords": coords, "label": label}


def save_slide_features(data_source: str, slide_id: str, features: torch.Tensor, coords=None, use_h5: bool = False) -> str:
    """Write one slide in the layout ``WSIClassificationDataset`` reads."""
    f = features.detach().to("cpu", torch.float32).contiguous()
    if use_h5:
        import h5py
        os.makedirs(os.path.join(data_source, "h5_files"), exist_ok=True)
        path = os.path.join(data_source, "h5_files", slide_id + ".h5")
        with h5py.File(path, "w") as h:
            h.create_dataset("features", data=f.numpy())
            h.create_dataset("coords", data=np.asarray(coords if coords is not None else np.zeros((f.shape[0], 2), np.int64)))
    else:
        os.makedirs(os.path.join(data_source, "pt_files"), exist_ok=True)
        path = os.path.join(data_source, "pt_files", slide_id + ".pt")
        torch.save(f, path)
    return path


def _loop(model, classifier: torch.Tensor, dataloader, device, softmax: bool, want_targets: bool):
    m = _engine(model, classifier, device=device)
    cls_t = classifier.to(m._device, torch.float32).t().contiguous()          # [C, 768] rows, as keep_similarity wants
    logits_all, coords_all, targets_all = {}, {}, {}
    for idx, data in enumerate(dataloader):                                   # batch size is always 1 slide
        feats = _normalized(m, data["features"])
        coords = data["coords"]
        if not isinstance(coords, list):
            coords = coords.squeeze(0).numpy()
        slide_id = dataloader.dataset.get_ids(idx)
        coords_all[slide_id] = coords
        logits_all[slide_id] = m.similarity(feats, cls_t, scale=10.0 if softmax else 1.0, mode="softmax" if softmax else "raw")
        if want_targets:
            t = data["label"]
            targets_all[slide_id] = t.item() if hasattr(t, "item") else t
    return logits_all, coords_all, targets_all


@torch.no_grad()
def run_subtyping(classifier, dataloader, device, model=None):
    """``run`` of subtyping_utils.py:12-35 -> (raw cosine logits per slide, coords, targets)."""
    return _loop(model, classifier, dataloader, device, softmax=False, want_targets=True)


@torch.no_grad()
def run_detection(classifier, dataloader, device, model=None):
    """``run`` of detection_utils.py:12-36 -> (softmax(10*logits) per slide, coords, targets)."""
    return _loop(model, classifier, dataloader, device, softmax=True, want_targets=True)


@torch.no_grad()
def run_segmentation(classifier, dataloader, device, model=None):
    """``run`` of segment_utils.py:16-42 -> (softmax(10*logits) per slide, coords)."""
    l, c, _ = _loop(model, classifier, dataloader, device, softmax=True, want_targets=False)
    return l, c


WSI_Classification_Dataset = WSIClassificationDataset        # the reference's spelling (utils.py:11)

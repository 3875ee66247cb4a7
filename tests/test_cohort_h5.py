"""The `.h5` branch of cohort mode (CLAM feature files: datasets `features` [N,768] f32 and `coords` [N,2];
WSI_evaluation/utils.py:50-55).  h5py is not installed in this image, so the branch is executed here against a minimal
stand-in module with the h5py calls the code uses (File as a context manager, create_dataset, `f[name][:]`), backed by
.npz files; with a real h5py present the second test runs the same round trip through actual HDF5."""
import sys
import types

import numpy as np
import pytest
import torch

from keep_amd import cohort


class _FakeFile:
    def __init__(self, path, mode="r"):
        self.path, self.mode, self.data = path, mode, {}
        if mode == "r":
            with np.load(path + ".npz") as z:
                self.data = {k: z[k] for k in z.files}

    def create_dataset(self, name, data):
        assert self.mode == "w"
        self.data[name] = np.asarray(data)

    def __getitem__(self, name):
        return self.data[name]

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if self.mode == "w":
            np.savez(self.path + ".npz", **self.data)
        return False


def _round_trip(tmp_path):
    g = torch.Generator().manual_seed(3)
    feats = torch.randn(41, 768, generator=g)
    coords = np.stack([np.arange(41) * 256, np.arange(41)[::-1] * 256], 1).astype(np.int64)
    cohort.save_slide_features(str(tmp_path), "s0", feats, coords, use_h5=True)
    cohort.save_slide_features(str(tmp_path), "s1", feats[:1], None, use_h5=True)            # no coordinates given: zeros
    rows = [{"slide_id": "s0", "Diagnosis": "B"}, {"slide_id": "s1", "Diagnosis": "A"}]
    ds = cohort.WSIClassificationDataset(rows, str(tmp_path), use_h5=True, label_map={"A": 0, "B": 1})
    assert len(ds) == 2
    item = ds[0]
    assert torch.equal(item["features"], feats) and item["features"].dtype == torch.float32
    assert np.array_equal(item["coords"].numpy(), coords) and item["label"] == 1
    assert ds[1]["features"].shape == (1, 768) and ds[1]["coords"].shape == (1, 2) and ds[1]["label"] == 0


def test_h5_branch_with_stand_in_module(tmp_path, monkeypatch):
    fake = types.ModuleType("h5py")
    fake.File = _FakeFile
    monkeypatch.setitem(sys.modules, "h5py", fake)
    _round_trip(tmp_path)


def test_h5_branch_with_real_h5py(tmp_path):
    pytest.importorskip("h5py")
    _round_trip(tmp_path)

"""Cohort mode (feature files + run() loops, WSI_evaluation/utils.py:11-61 and the three `run` functions)."""
import numpy as np
import pytest
import torch

from keep_amd import KEEPModel, cohort
from oracle import keep_oracle as O

pytestmark = pytest.mark.gpu


def test_pt_cohort_round_trip(tmp_path):
    g = torch.Generator().manual_seed(0)
    rows, feats = [], {}
    for i, n in enumerate((37, 1, 260)):
        sid = f"slide_{i}"
        feats[sid] = torch.randn(n, 768, generator=g) * 2.0
        cohort.save_slide_features(str(tmp_path), sid, feats[sid])
        rows.append({"slide_id": sid, "Diagnosis": ("A", "B", "A")[i]})
    ds = cohort.WSIClassificationDataset(rows, str(tmp_path), use_h5=False, label_map={"A": 0, "B": 1})
    dl = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False)
    cls = torch.nn.functional.normalize(torch.randn(768, 3, generator=g), dim=0)
    m = KEEPModel()
    logits, coords, targets = cohort.run_subtyping(m, cls, dl)
    probs, _, targets2 = cohort.run_detection(m, cls, dl)
    seg, _ = cohort.run_segmentation(m, cls, dl)
    assert targets == {"slide_0": 0, "slide_1": 1, "slide_2": 0} == targets2
    for sid, f in feats.items():
        ref = O.l2_normalize(f) @ cls
        assert (logits[sid].cpu() - ref).abs().max() < 2e-6
        assert (probs[sid].cpu() - O.sim_softmax(ref, 10.0)).abs().max() < 2e-6
        assert torch.equal(seg[sid], probs[sid]) and coords[sid] == []
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError):
            cohort.WSIClassificationDataset(rows, str(tmp_path), use_h5=True)[0]

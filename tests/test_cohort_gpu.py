"""Cohort mode (feature files + the three `run(classifier, dataloader, device)` loops, WSI_evaluation/utils.py:11-61,
subtyping_utils.py:12, detection_utils.py:12, segment_utils.py:16) against what the reference's own `run` functions returned
over the reference's own dataset class (tests/golden/wsi_callers.npz, tools/make_golden.py wsi_callers)."""
import os

import numpy as np
import pytest
import torch

from keep_amd import cohort
from keep_amd.wsi_evaluation import detection_utils, segment_utils, subtyping_utils
from keep_amd.wsi_evaluation.utils import WSI_Classification_Dataset
from oracle import keep_oracle as O
from standins import CALLER_DIAG_MAP, CALLER_SLIDES, caller_slide_features

pytestmark = pytest.mark.gpu


def test_run_loops_match_the_reference_run_functions(tmp_path, golden_dir):
    g = np.load(os.path.join(golden_dir, "wsi_callers.npz"))
    feats = caller_slide_features(int(g["feat_seed"]))
    assert abs(sum(float(f.double().abs().sum()) for f in feats.values()) - float(g["feats_checksum"])) < 1e-6 * float(g["feats_checksum"])
    rows = []
    for sid, _, diag in CALLER_SLIDES:
        cohort.save_slide_features(str(tmp_path), sid, feats[sid])
        rows.append({"slide_id": sid, "Diagnosis": diag})
    try:
        import pandas as pd
        df = pd.DataFrame(rows)                   # what the reference passes (utils.py:33: self.data.loc[ids, col])
    except ImportError:
        df = rows
    ds = WSI_Classification_Dataset(df, str(tmp_path), use_h5=False, label_map=CALLER_DIAG_MAP)
    dl = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False)
    device = "cuda:0"
    cls3 = torch.from_numpy(g["cls3"]).to(device)
    cls2 = cls3[:, :2].contiguous()
    logits, coords, targets = subtyping_utils.run(cls3, dl, device)
    probs, _, targets2 = detection_utils.run(cls2, dl, device)
    seg, seg_coords = segment_utils.run(cls2, dl, device)
    want_t = {sid: CALLER_DIAG_MAP[d] for sid, _, d in CALLER_SLIDES}
    assert targets == want_t == targets2 and list(seg_coords) == [sid for sid, _, _ in CALLER_SLIDES]
    for sid, f in feats.items():
        assert logits[sid].device.type == "cuda" and coords[sid] == []
        assert np.abs(logits[sid].cpu().numpy() - g[f"run_sub_{sid}"]).max() < 2e-6
        assert np.abs(probs[sid].cpu().numpy() - g[f"run_det_{sid}"]).max() < 2e-6
        assert np.abs(seg[sid].cpu().numpy() - g[f"run_seg_{sid}"]).max() < 2e-6
        assert (logits[sid].cpu() - O.l2_normalize(f) @ cls3.cpu()).abs().max() < 2e-6
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError):
            cohort.WSIClassificationDataset(rows, str(tmp_path), use_h5=True)[0]

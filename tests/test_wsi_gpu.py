"""Slide-level zero-shot logic on the GPU (keep_amd.wsi, called with the REFERENCE's signatures) against vectors produced by
the reference's own WSI_evaluation/*.py functions (tests/golden/wsi_logic.npz, wsi_callers.npz; tools/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from keep_amd import KEEPModel, wsi
from keep_amd.config import small_shape
from keep_amd.synth import synth_state_dict
from oracle import keep_oracle as O
from standins import CALLER_LABEL_MAP, CALLER_PROMPTS, HashTokenizer

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "wsi_logic.npz"))


@pytest.fixture(scope="module")
def model():
    return KEEPModel()          # similarity / refine kernels need no weights


def test_prompt_scores_and_selection(g, model):
    feats = torch.from_numpy(g["feats"])
    cls4 = [torch.from_numpy(c) for c in g["cls4"]]
    cls2 = [c[:, :2].contiguous() for c in cls4]
    scores = wsi.prompt_scores(feats, cls4, model=model).cpu().numpy()
    assert np.abs(scores - g["scores4"]).max() < 2e-6
    assert abs(wsi.rank_cls_score(O.l2_normalize(feats) @ cls4[3]) - float(g["scores4"][3])) < 2e-6
    ens4 = wsi.zero_shot_prompt_select(cls4, feats[None], int(g["topn"]), "cuda:0")       # [1,N,768] like a DataLoader batch
    ens2 = wsi.zero_shot_prompt_select(cls2, feats, topn=int(g["topn"]), device="cuda:0")
    assert np.abs(ens4.cpu().numpy() - g["ens4"]).max() < 1e-6
    assert np.abs(ens2.cpu().numpy() - g["ens2"]).max() < 1e-6


def test_fused_screening_modes_agree(g):
    """keep_prompt_scores: fused compensated GEMM (default), fused 3-pass GEMM and the unfused fp32 path (logits through HBM)
    give the reference's scores; C = 4 and C = 2; N not a multiple of the 256-row tile; K*C not a multiple of 256."""
    feats = torch.from_numpy(g["feats"])
    cls4 = [torch.from_numpy(c) for c in g["cls4"]]
    cls2 = [c[:, :2].contiguous() for c in cls4]
    fn = O.l2_normalize(feats)
    ref2 = np.array([O.rank_cls_score(fn @ c) for c in cls2])
    got = {}
    for mode in (0, 1, 2):
        m = KEEPModel()
        m.set_option("fused_screening", mode)
        s4 = wsi.prompt_scores(feats, cls4, model=m).cpu().numpy()
        s2 = wsi.prompt_scores(feats[:517], cls2, model=m).cpu().numpy()
        r2 = np.array([O.rank_cls_score(fn[:517] @ c) for c in cls2])
        got[mode] = s4
        print(f"[fused_screening={mode}] max err C=4 {np.abs(s4 - g['scores4']).max():.2e}  C=2 {np.abs(s2 - r2).max():.2e}")
        assert np.abs(s4 - g["scores4"]).max() < 2e-6 and np.abs(s2 - r2).max() < 2e-6
    assert np.abs(got[1] - got[0]).max() < 2e-6


def test_subtyping_detection_segmentation(g, model):
    feats = torch.from_numpy(g["feats"])
    ens4, ens2 = torch.from_numpy(g["ens4"]), torch.from_numpy(g["ens2"])
    assert int(wsi.zero_shot_subtyping(ens4, feats, g["coords256"], patch_size=256, overlap=True)) == int(g["sub_label"])
    probs = model.similarity(O.l2_normalize(feats), ens4.t().contiguous(), scale=10.0, mode="softmax")
    coords, mean, idx = wsi.refine(probs, g["coords256"], 256, True)
    assert np.array_equal(coords.cpu().numpy(), g["sub_keys"])
    assert np.array_equal(mean.argmax(1).cpu().numpy(), g["sub_preds"])
    assert wsi.zero_shot_detection(ens2, feats, g["coords256"], patch_size=256, overlap=False) == pytest.approx(float(g["det_ratio"]), abs=1e-12)
    assert wsi.zero_shot_detection(ens2.cuda(), feats.cuda(), g["coords256"], 256, True) == pytest.approx(float(g["det_ratio_overlap"]), abs=1e-12)
    seg = wsi.zero_shot_segment(ens2, feats, g["coords224"], None, patch_size=224, overlap=True)
    keys = np.array([[int(s) for s in k.split("_")] for k in seg])
    assert np.array_equal(keys, g["seg_keys"])
    assert np.abs(np.array(list(seg.values())) - g["seg_probs"]).max() < 1e-6


def test_refine_matches_oracle_on_random_grids(model):
    gen = torch.Generator().manual_seed(3)
    for n, grid, C, patch, overlap in ((1, 4, 2, 224, True), (257, 20, 4, 256, True), (1000, 25, 3, 256, False), (5000, 90, 2, 224, True)):
        cells = torch.randint(0, grid * grid, (n,), generator=gen)               # duplicates on purpose
        coords = torch.stack([(cells % grid) * patch + 17, (cells // grid) * patch - 5], dim=1).numpy()
        probs = torch.softmax(torch.randn(n, C, generator=gen) * 3, dim=1)
        keys, mean = O.refine_mean_probs(probs, coords, patch, overlap)
        c, m, _ = wsi.refine(probs, coords, patch, overlap, model=model)
        assert np.array_equal(c.cpu().numpy(), np.array(keys))
        assert np.array_equal(m.cpu().numpy(), mean), "float32 neighbour means must be bit-identical to numpy's"


def test_classifier_builder_dedupes_prompt_strings():
    sd = synth_state_dict(small_shape(1, 2), seed=9, vision=False)
    m = KEEPModel(precision="strict", towers=("text",))
    m.load_state_dict(sd)
    m.to("cuda:0")
    KEEP_model = {"model": m, "tokenizer": HashTokenizer()}
    label_map = {"CHRCC": 0, "CCRCC": 1, "PRCC": 2}
    prompts = [{"classnames": {"CHRCC": "chromophobe renal cell carcinoma", "CCRCC": "clear cell renal cell carcinoma",
                               "PRCC": "papillary renal cell carcinoma", "Normal": "normal kidney tissue"},
                "templates": t} for t in ("an H&E image of CLASSNAME.", "a histopathology slide showing CLASSNAME.", "an H&E image of CLASSNAME.")]
    cache = wsi.TextEmbeddingCache(KEEP_model, "cuda:0")
    HashTokenizer.calls = 0
    cls = [wsi.get_zeroshot_classifier(KEEP_model, label_map, p, "cuda:0", add_normal=True, cache=cache) for p in prompts]
    assert HashTokenizer.calls == 8                      # 12 requests, 8 distinct strings
    bank = wsi.build_classifier_bank(KEEP_model, label_map, prompts, "cuda:0", add_normal=True, cache=cache)
    assert HashTokenizer.calls == 8 and len(bank) == 3
    for a, b in zip(bank, cls):
        assert a.shape == (768, 4) and (a - b).abs().max() < 1e-6
    as_json = {str(i): p for i, p in enumerate(prompts)}
    bank2 = wsi.build_classifier_bank(KEEP_model, label_map, as_json, "cuda:0", add_normal=True, cache=cache)
    assert all(torch.equal(a, b) for a, b in zip(bank, bank2))
    assert cls[0].shape == (768, 4) and torch.equal(cls[0], cls[2])
    # column c equals the (re-normalised) embedding of the filled template, as utils.py:69-83 builds it
    text = prompts[1]["templates"].replace("CLASSNAME", prompts[1]["classnames"]["PRCC"])
    with torch.no_grad():
        ref = O.encode_text(sd, HashTokenizer()([text]))[0]
    assert (cls[1][:, 2].cpu() - ref / ref.norm()).abs().max() < 5e-6


# ------------------------------------------------------------------ rows a9 / a10 vs the reference's own functions
@pytest.fixture(scope="module")
def callers(golden_dir):
    return np.load(os.path.join(golden_dir, "wsi_callers.npz"))


def test_classifier_builders_match_the_reference_functions(callers):
    """`zero_shot_classifier` / `get_zeroshot_classifier` imported under the reference's module name, called with the
    reference's arguments, against what the reference's own functions returned for the same stand-in tokenizer and the
    same seeded text-tower weights (tools/make_golden.py wsi_callers; text tower there = transformers.BertModel)."""
    import sys
    import keep_amd.wsi_evaluation as W
    sys.path.insert(0, W.PATH)
    try:
        for name in ("utils", "subtyping_utils", "detection_utils", "segment_utils"):
            sys.modules.pop(name, None)
        from utils import get_zeroshot_classifier, zero_shot_classifier            # the scripts' own import line (zeroshot_subtyping_WSI.py:3)
    finally:
        sys.path.remove(W.PATH)
    sd = synth_state_dict(small_shape(1, 2), seed=int(callers["weight_seed"]), vision=False)
    for precision, tol in (("comp", 5e-6), ("strict", 5e-6)):          # the text tower runs split products in both
        m = KEEPModel(precision=precision, towers=("text",))
        m.load_state_dict(sd)
        device = "cuda:0"
        m = m.to(device)
        m.eval()
        KEEP_model = {"model": m, "tokenizer": HashTokenizer()}
        for i, p in enumerate(CALLER_PROMPTS):
            got = get_zeroshot_classifier(KEEP_model, CALLER_LABEL_MAP, p, device, add_normal=True)
            assert got.shape == (768, 4) and got.device.type == "cuda"
            assert np.abs(got.cpu().numpy() - callers[f"cls_normal_{i}"]).max() < tol
            got = get_zeroshot_classifier(KEEP_model, CALLER_LABEL_MAP, p, device)
            assert np.abs(got.cpu().numpy() - callers[f"cls_plain_{i}"]).max() < tol
        names = ["lung adenocarcinoma", "normal tissue"]
        assert np.abs(zero_shot_classifier(KEEP_model, names, "an H&E image of CLASSNAME.", device).cpu().numpy() - callers["zsc_str"]).max() < tol
        assert np.abs(zero_shot_classifier(KEEP_model, names, ["CLASSNAME.", "a photo of CLASSNAME."], device).cpu().numpy() - callers["zsc_list"]).max() < tol
        # the per-model prompt cache is invisible: same answer with it switched off (every string embedded on every call, as the reference does)
        wsi.PROMPT_CACHE = False
        try:
            again = get_zeroshot_classifier(KEEP_model, CALLER_LABEL_MAP, CALLER_PROMPTS[0], device, add_normal=True)
        finally:
            wsi.PROMPT_CACHE = True
        assert torch.equal(again, get_zeroshot_classifier(KEEP_model, CALLER_LABEL_MAP, CALLER_PROMPTS[0], device, add_normal=True))


def test_refine_seg_dicts_have_the_reference_form(g, model):
    """The three `refine_seg` variants return what the reference's return: {"x_y": label}, ({"x_y": 0/1}, {"x_y": p}), {"x_y": p}."""
    from keep_amd.wsi_evaluation import detection_utils, segment_utils, subtyping_utils
    feats = torch.from_numpy(g["feats"])
    ens4, ens2 = torch.from_numpy(g["ens4"]), torch.from_numpy(g["ens2"])
    p4 = model.similarity(O.l2_normalize(feats), ens4.t().contiguous(), scale=10.0, mode="softmax")
    p2 = model.similarity(O.l2_normalize(feats), ens2.t().contiguous(), scale=10.0, mode="softmax")
    sub = subtyping_utils.refine_seg(p4, g["coords256"], patch_size=256, overlap=True)
    assert list(sub.keys()) == [f"{x}_{y}" for x, y in g["sub_keys"]] and list(sub.values()) == g["sub_preds"].tolist()
    preds, probs = detection_utils.refine_seg(p2, g["coords256"], patch_size=256, threshold=0.5, overlap=False)
    assert sum(preds.values()) / len(preds) == pytest.approx(float(g["det_ratio"]), abs=1e-12) and set(preds.values()) <= {0, 1}
    assert list(probs.keys()) == list(preds.keys())
    seg = segment_utils.refine_seg(p2, g["coords224"], patch_size=224, overlap=True)
    assert np.abs(np.array(list(seg.values())) - g["seg_probs"]).max() < 1e-6
    with pytest.raises(NotImplementedError):
        segment_utils.zero_shot_segment(ens2, feats, g["coords224"], "mask.tif")


def test_unscreened_ensemble_on_the_engine(g):
    """Row a13, the `prompt_screening = False` branch (zeroshot_subtyping_WSI.py:68-76): same seeded picks as the oracle's restatement, the
    column normalisation on the engine's row kernel; classifiers on the host come back on the host."""
    cls4 = [torch.from_numpy(c) for c in g["cls4"]]
    want = O.random_prompt_ensemble(cls4, 10)
    got = wsi.random_prompt_ensemble(cls4, 10)
    assert got.device.type == "cpu" and got.shape == want.shape and (got - want).abs().max() < 2e-7
    got = wsi.random_prompt_ensemble([c.cuda() for c in cls4], 10)
    assert got.device.type == "cuda" and (got.cpu() - want).abs().max() < 2e-7

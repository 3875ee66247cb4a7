"""Oracle (CPU restatement) against the committed golden vectors.

The vectors were produced in the build container by tools/make_golden.py from
transformers.BertModel (the reference's own text-tower class), transformers.Dinov2Model configured
as ViT-L/16 (independent implementation of the timm block arithmetic) and the reference's
WSI_evaluation/*utils.py.
"""
import os

import numpy as np
import pytest
import torch

from keep_amd.config import KEEPShape, small_shape
from keep_amd.synth import synth_state_dict, synth_tiles
from oracle import keep_oracle as O


def checksum(t):
    return float(t.double().abs().sum())


@pytest.mark.parametrize("depth", [2, 24])
def test_image_tower_matches_golden(golden_dir, depth):
    g = np.load(os.path.join(golden_dir, f"vit_d{depth}.npz"))
    shape = KEEPShape() if depth == 24 else small_shape(vit_depth=depth)
    sd = synth_state_dict(shape, seed=int(g["weight_seed"]), text=False)
    x = synth_tiles(int(g["batch"]), seed=int(g["tile_seed"]))
    assert checksum(x) == pytest.approx(float(g["tiles_checksum"]), rel=1e-12), "seeded tile generator drifted"
    assert checksum(sd["visual.blocks.0.attn.qkv.weight"]) == pytest.approx(float(g["qkv0_checksum"]), rel=1e-12)
    with torch.no_grad():
        cls = O.vit_forward(sd, x, depth)
        feat = O.encode_image(sd, x)
    assert np.abs(cls.numpy() - g["cls"]).max() < 2e-4          # LN'd CLS token, |values| ~ 1
    assert np.abs(feat.numpy() - g["features"]).max() < 2e-6    # unit-norm 768-d features (pin 1: transformers' Dinov2Model as ViT-L/16)
    # pin 2: timm's module tree on the ATen ops timm dispatches (tools/make_golden.py `_TimmViT`), loaded strictly from the release key layout
    assert np.abs(feat.numpy() - g["features_aten_timm"]).max() < 1e-6
    assert float(g["pins_dfeat"]) < 1e-6 and float(g["oracle_dfeat_aten_timm"]) < 1e-6
    # pin 3, present once tools/pin_against_timm.py has run somewhere timm is installed: timm's own vit_large_patch16_224 built as the reference does
    if "features_timm" in g.files:
        assert np.abs(feat.numpy() - g["features_timm"]).max() < 1e-6
    assert np.allclose(np.linalg.norm(feat.numpy(), axis=1), 1.0, atol=1e-6)


@pytest.mark.parametrize("layers", [2, 12])
def test_text_tower_matches_golden(golden_dir, layers):
    g = np.load(os.path.join(golden_dir, f"bert_l{layers}.npz"))
    shape = KEEPShape() if layers == 12 else small_shape(bert_layers=layers)
    sd = synth_state_dict(shape, seed=int(g["weight_seed"]), vision=False)
    assert checksum(sd["text.embeddings.word_embeddings.weight"]) == pytest.approx(float(g["word_emb_checksum"]), rel=1e-12)
    toks = {k: torch.from_numpy(g[k].astype(np.int64)) for k in ("input_ids", "token_type_ids", "attention_mask")}
    with torch.no_grad():
        feat = O.encode_text(sd, toks)
    assert np.abs(feat.numpy() - g["features"]).max() < 2e-6


def test_padded_equals_truncated_text(golden_dir):
    """masked keys get exactly zero weight, so truncating to the valid length is exact."""
    g = np.load(os.path.join(golden_dir, "bert_l2.npz"))
    sd = synth_state_dict(small_shape(bert_layers=2), seed=int(g["weight_seed"]), vision=False)
    ids = torch.from_numpy(g["input_ids"].astype(np.int64))[2:3]
    mask = torch.from_numpy(g["attention_mask"].astype(np.int64))[2:3]
    L = int(mask.sum())
    assert 0 < L < 256 and bool(mask[0, :L].all())
    with torch.no_grad():
        full = O.encode_text(sd, {"input_ids": ids, "attention_mask": mask})
        cut = O.encode_text(sd, {"input_ids": ids[:, :L], "attention_mask": mask[:, :L]})
    assert (full - cut).abs().max() < 2e-6


def test_wsi_host_logic_matches_reference_outputs(golden_dir):
    g = np.load(os.path.join(golden_dir, "wsi_logic.npz"))
    feats = torch.from_numpy(g["feats"])
    cls4 = [torch.from_numpy(c) for c in g["cls4"]]
    cls2 = [c[:, :2].contiguous() for c in cls4]
    topn = int(g["topn"])
    fn = O.l2_normalize(feats)
    scores = np.array([O.rank_cls_score(fn @ c) for c in cls4])
    assert np.abs(scores - g["scores4"]).max() < 1e-6
    ens4 = O.zero_shot_prompt_select(cls4, feats, topn)
    ens2 = O.zero_shot_prompt_select(cls2, feats, topn)
    assert np.abs(ens4.numpy() - g["ens4"]).max() < 1e-6
    assert np.abs(ens2.numpy() - g["ens2"]).max() < 1e-6
    ens4g, ens2g = torch.from_numpy(g["ens4"]), torch.from_numpy(g["ens2"])
    assert O.zero_shot_subtyping(ens4g, feats, g["coords256"], 256, True) == int(g["sub_label"])
    keys, mean = O.refine_mean_probs(O.sim_softmax(fn @ ens4g, 10.0), g["coords256"], 256, True)
    assert np.array_equal(np.array(keys), g["sub_keys"])
    assert np.array_equal(mean.argmax(1), g["sub_preds"])
    assert O.zero_shot_detection(ens2g, feats, g["coords256"], 256, False) == pytest.approx(float(g["det_ratio"]), abs=1e-12)
    assert O.zero_shot_detection(ens2g, feats, g["coords256"], 256, True) == pytest.approx(float(g["det_ratio_overlap"]), abs=1e-12)
    keys, probs = O.zero_shot_segment_probs(ens2g, feats, g["coords224"], 224, True)
    assert np.array_equal(np.array(keys), g["seg_keys"])
    assert np.abs(probs - g["seg_probs"]).max() < 1e-6


def test_tile_eval_protocol_matches_reference_outputs(golden_dir):
    """oracle restatement of training/path_training/zero_shot.py vs the result dict the reference itself produced."""
    import json
    g = np.load(os.path.join(golden_dir, "tile_eval.npz"))
    names = [str(n) for n in g["names"]]
    labels = [str(x) for x in g["labels"]]
    val = O.tile_classification_rounds(g["img"], {n: g["caps"][i] for i, n in enumerate(names)}, labels)
    assert np.abs(val - g["wf1_rounds"]).max() < 1e-12
    q = O.wf1_quartiles(val)
    assert abs(q["zeroshot-cls-WF1-median"] - float(g["wf1_median"])) < 1e-12
    assert abs(q["zeroshot-cls-WF1-Q1"] - float(g["wf1_q1"])) < 1e-12 and abs(q["zeroshot-cls-WF1-Q3"] - float(g["wf1_q3"])) < 1e-12
    r = O.retrieval_p_at_k(g["ret_img"], g["ret_txt"])
    assert r["p@10"] == float(g["p10"]) and r["p@50"] == float(g["p50"])
    caps = O.label2cap(json.loads(str(g["prompts"])))
    assert list(caps) == names and all(len(v) == 50 for v in caps.values())


def test_weighted_f1_is_sklearns():
    """zeroshot_metrics.py:31 calls sklearn; the restatement and the confusion-matrix form used on the GPU side agree
    with it, including labels that are only predicted or only true."""
    from sklearn.metrics import f1_score
    from keep_amd.tile_eval import weighted_f1_from_confusion
    rng = np.random.default_rng(3)
    for trial in range(20):
        n, c = int(rng.integers(5, 200)), int(rng.integers(2, 7))
        yt = rng.integers(0, c, n); yp = rng.integers(0, c + (trial % 2), n)
        if trial % 3 == 0:
            yt[yt == 0] = 1                                   # class 0 only predicted
        ref = f1_score(yt, yp, average="weighted", zero_division=0)
        assert abs(O.weighted_f1(list(yt), list(yp)) - ref) < 1e-12
        A = c + 1
        conf = np.bincount(yt * A + yp, minlength=A * A).reshape(A, A)
        assert abs(weighted_f1_from_confusion(conf) - ref) < 1e-12


def test_unscreened_ensemble_is_deterministic(golden_dir):
    """`prompt_screening = False` branch (zeroshot_subtyping_WSI.py:68-76): the seeded picks are fixed numbers."""
    import random
    g = np.load(os.path.join(golden_dir, "wsi_logic.npz"))
    cls4 = [torch.from_numpy(c) for c in g["cls4"]]
    picks = []
    for c in range(10):
        random.seed(c)
        picks.append(random.randint(0, len(cls4) - 1))
    want = torch.nn.functional.normalize(sum(cls4[i] for i in picks), p=2, dim=0)
    assert torch.allclose(O.random_prompt_ensemble(cls4, 10), want, atol=1e-7)
    # (the product's random_prompt_ensemble normalises on the engine: tests/test_wsi_gpu.py::test_unscreened_ensemble_on_the_engine)
    assert (want.norm(dim=0) - 1).abs().max() < 1e-6


def test_flop_model_matches_survey():
    from keep_amd.config import bert_flops_per_prompt, vit_flops_per_tile
    assert vit_flops_per_tile() == 123_110_129_664
    assert bert_flops_per_prompt() == 45_903_642_624

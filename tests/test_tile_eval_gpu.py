"""Tile-level zero-shot evaluation (keep_amd.tile_eval) against the result dict of the reference's own
training/path_training/zero_shot.py::zero_shot_eval (tests/golden/tile_eval.npz, tools/make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

from keep_amd import KEEPModel, tile_eval
from keep_amd.config import small_shape
from keep_amd.synth import synth_state_dict, synth_tiles
from oracle import keep_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "tile_eval.npz"))


def test_classification_rounds_match_reference(g):
    m = KEEPModel()                                         # the evaluation kernels need no weights
    names = [str(n) for n in g["names"]]
    caps = {n: torch.from_numpy(g["caps"][i]) for i, n in enumerate(names)}
    val = tile_eval.classification_rounds(m, torch.from_numpy(g["img"]), caps, [str(x) for x in g["labels"]])
    assert val.shape == (50,) and np.abs(val - g["wf1_rounds"]).max() < 1e-12
    q1, med, q3 = np.percentile(val, (25, 50, 75), method="midpoint")
    assert abs(med - float(g["wf1_median"])) < 1e-12 and abs(q1 - float(g["wf1_q1"])) < 1e-12 and abs(q3 - float(g["wf1_q3"])) < 1e-12
    # a label no caption class covers is never predicted: same as sklearn's union-of-labels weighting
    lab = [str(x) for x in g["labels"]]
    lab[5] = "Unseen"
    val2 = tile_eval.classification_rounds(m, torch.from_numpy(g["img"]), caps, lab)
    ref2 = O.tile_classification_rounds(g["img"], {n: g["caps"][i] for i, n in enumerate(names)}, lab)
    assert np.abs(val2 - ref2).max() < 1e-12


def test_retrieval_matches_reference(g):
    m = KEEPModel()
    r = tile_eval.retrieval_metrics(m, torch.from_numpy(g["ret_img"]), torch.from_numpy(g["ret_txt"]))
    assert r["p@10"] == float(g["p10"]) and r["p@50"] == float(g["p50"])
    # explicit targets, fewer captions than images, and a tie: equal scores list the higher index first
    img = torch.nn.functional.normalize(torch.from_numpy(g["ret_img"][:40]), dim=-1)
    img[7] = img[3]
    txt = img[[3, 7, 11]].clone()
    rank = tile_eval.retrieval_ranks(m, img, txt, targets=torch.tensor([3, 7, 11])).cpu().tolist()
    assert rank == [1, 0, 0]
    for t, tb in enumerate(txt.numpy()):
        order = list(np.argsort(tb.dot(img.numpy().T), kind="stable")[::-1])
        assert order.index([3, 7, 11][t]) == rank[t]


class _Tok:
    def __call__(self, texts, add_special_tokens=True, max_length=256, padding="max_length", truncation=True, return_tensors="pt"):
        ids = torch.zeros(len(texts), max_length, dtype=torch.int64)
        mask = torch.zeros_like(ids)
        for i, t in enumerate(texts):
            toks = [2] + [5 + (sum(map(ord, w)) * 31 + len(w)) % 30000 for w in t.lower().split()][: max_length - 2] + [3]
            ids[i, :len(toks)] = torch.tensor(toks)
            mask[i, :len(toks)] = 1
        return {"input_ids": ids, "token_type_ids": torch.zeros_like(ids), "attention_mask": mask}


def test_zero_shot_eval_end_to_end(g):
    """Encoders + protocol together on a depth-2 model: the engine's result dict equals the oracle protocol run on the
    oracle's own embeddings (strict mode, so no argmax can flip)."""
    shape = small_shape(2, 2)
    sd = synth_state_dict(shape, seed=61)
    m = KEEPModel(shape, precision="strict")
    m.load_state_dict(sd)
    m.to("cuda:0")
    prompts = json.loads(str(g["prompts"]))
    names = list(prompts["0"]["classnames"].keys())
    tiles = synth_tiles(24, seed=62)
    labels = [names[i % 4] for i in range(24)]
    texts = [f"caption number {i} of tissue {i * 7 % 5}" for i in range(24)]
    data = {"zeroshot_cls": [(tiles[i:i + 8], labels[i:i + 8]) for i in range(0, 24, 8)],
            "zeroshot_ret": [(tiles[i:i + 12], texts[i:i + 12]) for i in range(0, 24, 12)]}
    res = tile_eval.zero_shot_eval(m, _Tok(), data, prompts)
    with torch.no_grad():
        oi = O.encode_image(sd, tiles)
        caps = {n: O.encode_text(sd, _Tok()(c)).numpy() for n, c in O.label2cap(prompts).items()}
        ot = O.encode_text(sd, _Tok()(texts))
    ref = O.wf1_quartiles(O.tile_classification_rounds(oi.numpy(), caps, labels))
    ret = O.retrieval_p_at_k(oi.numpy(), ot.numpy())
    for k, v in ref.items():
        assert abs(res[k] - v) < 1e-12, (k, res[k], v)
    assert res["zeroshot-ret-p@10"] == ret["p@10"] and res["zeroshot-ret-p@50"] == ret["p@50"]
    assert tile_eval.label2cap(prompts) == O.label2cap(prompts)

"""Row a1 on the GPU: a release directory (config.json + weights file + vocab.txt) opened the way the reference scripts open theirs
(``AutoModel.from_pretrained(model_path, ...).to(device)``, zeroshot_subtyping_WSI.py:44; ``AutoConfig`` + ``from_config`` +
``load_state_dict(strict=True)``, keep_inference.py:80-83) and then driven through the subtyping script's call sequence -- written here in
this repository's own words, with every stage compared with the CPU oracle."""
import json
import os

import numpy as np
import pytest
import torch

import keep_amd.hf                                           # noqa: F401
from keep_amd import KEEPModel
from keep_amd.config import small_shape
from keep_amd.synth import synth_state_dict, synth_tiles, synthetic_rcc_prompts, write_synthetic_release
from keep_amd.tokenizer import load_tokenizer
from oracle import keep_oracle as O
from transformers import AutoConfig, AutoModel

pytestmark = pytest.mark.gpu
SHAPE = small_shape(2, 2)
SEED = 6


@pytest.fixture(scope="module")
def release(tmp_path_factory):
    return write_synthetic_release(str(tmp_path_factory.mktemp("KEEP_release")), SHAPE, seed=SEED)


@pytest.fixture(scope="module")
def sd():
    return synth_state_dict(SHAPE, seed=SEED)


def test_from_pretrained_directory_runs_and_matches_the_oracle(release, sd):
    model = AutoModel.from_pretrained(release, trust_remote_code=False).to("cuda:0")
    assert type(model) is KEEPModel and model.eval() is model and model.device == torch.device("cuda", 0)
    assert model.calibration is not None and model.calibration["precision"] in ("comp", "strict")      # load ends with calibrate()
    tok = load_tokenizer(release)
    texts = ["an H&E image of clear cell renal cell carcinoma.", "normal kidney tissue", "a histopathology slide showing tumor"]
    enc = tok(texts, max_length=256, padding="max_length", truncation=True, return_tensors="pt")
    x = synth_tiles(3, seed=21)
    out = model(x.to("cuda:0"), enc.to("cuda:0"))                        # forward: keep_inference.py:65-73
    assert set(out) == {"vision_features", "text_features"}
    with torch.no_grad():
        wi, wt = O.encode_image(sd, x), O.encode_text(sd, {k: v for k, v in enc.items()})
    sim = (out["vision_features"] @ out["text_features"].T).cpu()        # keep_inference.py:104
    assert tuple(sim.shape) == (3, 3)
    assert (out["text_features"].cpu() - wt).abs().max() < 5e-6
    assert (sim - wi @ wt.T).abs().max() < 1e-4                          # north-star tolerance on the cosines
    # host tensors in -> host tensors out (the quick start keeps everything on the CPU)
    assert model.encode_image(x).device.type == "cpu" and model.encode_text(enc).device.type == "cpu"


def test_config_then_state_dict_strict(release, sd, tmp_path):
    """keep_inference.py:80-83 with pytorch_model.bin, and what strict=True means for a missing / unexpected key."""
    rel = write_synthetic_release(str(tmp_path / "bin"), SHAPE, seed=SEED, weights="bin")
    config = AutoConfig.from_pretrained(os.path.join(rel, "config.json"))
    model = AutoModel.from_config(config)
    state_dict = torch.load(os.path.join(rel, "pytorch_model.bin"), map_location="cpu")
    model.load_state_dict(state_dict, strict=True)
    model.to("cuda:0").eval()
    x = synth_tiles(2, seed=22)
    a = model.encode_image(x.cuda())
    b = AutoModel.from_pretrained(release).to("cuda:0").encode_image(x.cuda())
    assert torch.equal(a, b)                                             # .bin and .safetensors hold the same tensors
    missing = {k: v for k, v in state_dict.items() if k != "visual.blocks.1.mlp.fc1.bias"}
    with pytest.raises(RuntimeError, match="visual.blocks.1.mlp.fc1.bias"):
        AutoModel.from_config(config).load_state_dict(missing, strict=True).to("cuda:0")
    extra = dict(state_dict, **{"visual.blocks.0.attn.q_norm.weight": torch.ones(64)})
    with pytest.raises(RuntimeError, match="q_norm"):
        AutoModel.from_config(config).load_state_dict(extra, strict=True).to("cuda:0")
    AutoModel.from_config(config).load_state_dict(extra, strict=False).to("cuda:0")      # tolerated when not strict
    dropped = write_synthetic_release(str(tmp_path / "dropped"), SHAPE, seed=SEED, drop_keys=("text.pooler.dense.weight",))
    with pytest.raises(RuntimeError, match="text.pooler.dense.weight"):
        AutoModel.from_pretrained(dropped).to("cuda:0")


def test_subtyping_flow_from_a_release_directory(release, sd, tmp_path):
    """What WSI_evaluation/zeroshot_subtyping_WSI.py does after its constants: open model + tokenizer from one directory, build one
    classifier per prompt set (add_normal=True), screen them on the slide's tile features, keep the top-n, refine on the tile grid
    and report the slide label -- here on a synthetic release, a synthetic prompt file and a synthetic slide, each stage against the
    CPU oracle's restatement of the same reference function."""
    from keep_amd.wsi_evaluation import subtyping_utils, utils
    device = "cuda:0"
    prompt_file = tmp_path / "prompts.json"
    json.dump(synthetic_rcc_prompts(12), open(prompt_file, "w"))
    prompts = json.load(open(prompt_file))
    label_map, topn = {"CHRCC": 0, "CCRCC": 1, "PRCC": 2}, 5
    engine = AutoModel.from_pretrained(release).to(device).eval()
    bundle = {"model": engine, "tokenizer": load_tokenizer(release)}
    n_tiles = 300
    tiles = synth_tiles(n_tiles, seed=23)
    feats = torch.cat([engine.encode_image(tiles[i:i + 128].to(device)) for i in range(0, n_tiles, 128)])
    side = 18
    coords = np.stack([(np.arange(n_tiles) % side) * 256, (np.arange(n_tiles) // side) * 256], 1)
    bank = [utils.get_zeroshot_classifier(bundle, label_map, prompts[str(i)], device, add_normal=True) for i in range(len(prompts))]
    chosen = utils.zero_shot_prompt_select(bank, feats, topn=topn, device=device)
    label = subtyping_utils.zero_shot_subtyping(chosen, feats, coords, patch_size=256, overlap=True)

    # the oracle's side: text embeddings of the same strings -> classifiers -> screening -> label, all fp32 on the CPU
    tok = bundle["tokenizer"]
    order = ["CHRCC", "CCRCC", "PRCC", "Normal"]
    with torch.no_grad():
        want_bank = []
        for i in range(len(prompts)):
            p = prompts[str(i)]
            texts = [p["templates"].replace("CLASSNAME", p["classnames"][c]) for c in order]
            emb = O.encode_text(sd, dict(tok(texts, max_length=256, padding="max_length", truncation=True, return_tensors="pt")))
            want_bank.append(O.build_classifier(emb))
        want_feats = O.encode_image(sd, tiles)
        want_chosen = O.zero_shot_prompt_select(want_bank, want_feats, topn)
        want_label = O.zero_shot_subtyping(want_chosen, want_feats, coords, 256, True)
    for got, want in zip(bank, want_bank):
        assert got.shape == (768, 4) and (got.cpu() - want).abs().max() < 5e-6
    assert ((feats.cpu() @ want_chosen) - (want_feats @ want_chosen)).abs().max() < 1e-4
    assert (chosen.cpu() - want_chosen).abs().max() < 2e-5                 # same top-n picks, same sum
    assert int(label) == int(want_label)

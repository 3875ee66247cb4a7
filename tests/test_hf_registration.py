"""Row a1 of SURVEY.md section 8: ``KEEPConfig`` / ``config.json`` / ``Auto*`` registration (keep_inference.py:9-22, 75-81;
zeroshot_subtyping_WSI.py:44).  Runs without a GPU: constructing the engine object allocates nothing on a device (weights are
repacked at ``.to('cuda')``), so the registration, the config parsing and the error behaviour are checked here; the same directory is
opened and RUN on the GPU in tests/test_release_dir_gpu.py."""
import json
import os

import pytest
import torch

import keep_amd.hf                                          # noqa: F401  -- the two register() calls of keep_inference.py:75-76
from keep_amd import KEEPModel
from keep_amd.config import KEEPShape, small_shape
from keep_amd.hf import KEEPConfig
from keep_amd.synth import write_synthetic_release
from transformers import AutoConfig, AutoModel


@pytest.fixture(scope="module")
def release(tmp_path_factory):
    return write_synthetic_release(str(tmp_path_factory.mktemp("KEEP_release")), small_shape(1, 1), seed=4)


def test_autoconfig_reads_the_release_config(release):
    """``AutoConfig.from_pretrained(model_path + 'config.json')`` -- keep_inference.py:80."""
    cfg = AutoConfig.from_pretrained(os.path.join(release, "config.json"))
    assert isinstance(cfg, KEEPConfig) and cfg.model_type == "keep"
    assert cfg.projection_dim == 768 and cfg.text_config["num_hidden_layers"] == 1 and cfg.vision_config is None
    shape = cfg.to_shape()
    assert shape.text.num_hidden_layers == 1 and shape.text.vocab_size == 30522 and shape.vision.depth == 24
    assert AutoConfig.from_pretrained(release).to_dict()["text_config"] == cfg.text_config        # the directory form


def test_automodel_from_config_is_the_engine(release):
    """``AutoModel.from_config(config)`` -- keep_inference.py:81: the engine class, no weights yet, the reference's attributes."""
    cfg = AutoConfig.from_pretrained(os.path.join(release, "config.json"))
    model = AutoModel.from_config(cfg)
    assert type(model) is KEEPModel and model.hf_config is cfg
    assert model.config.text.num_hidden_layers == 1 and model.config.projection_dim == 768
    assert model.eval() is model and model.to("cpu") is model
    assert abs(float(model.logit_scale) - 3.2188758) < 1e-6                  # log(1 / 0.04), keep_inference.py:52
    with pytest.raises(Exception, match="no weights loaded|no GPU"):
        model.encode_image(torch.zeros(1, 3, 224, 224))


def test_bad_text_config_raises(tmp_path):
    """Only what the hot path implements is accepted: erf-GELU, absolute positions."""
    for field, value, what in (("hidden_act", "relu", "erf-GELU"), ("position_embedding_type", "relative_key", "absolute")):
        cfg = KEEPConfig(text_config={"hidden_act": "gelu", field: value}, projection_dim=768)
        with pytest.raises(ValueError, match=what):
            AutoModel.from_config(cfg)
        d = tmp_path / f"bad_{field}"
        d.mkdir()
        json.dump({"model_type": "keep", "projection_dim": 768, "text_config": {field: value}}, open(d / "config.json", "w"))
        with pytest.raises(ValueError, match=what):
            KEEPShape.from_config_json(str(d / "config.json"))


def test_config_round_trip_and_defaults(tmp_path):
    cfg = KEEPConfig(vision_config={"anything": 1}, text_config={"num_hidden_layers": 3, "intermediate_size": 3072}, projection_dim=768)
    cfg.save_pretrained(str(tmp_path))
    back = AutoConfig.from_pretrained(str(tmp_path))
    assert isinstance(back, KEEPConfig) and back.text_config["num_hidden_layers"] == 3 and back.vision_config == {"anything": 1}
    assert KEEPShape.from_config_json({}).text.num_hidden_layers == 12        # PubMedBERT-base defaults
    assert KEEPModel(str(tmp_path / "config.json")).config.text.num_hidden_layers == 3


def test_from_pretrained_needs_a_local_release(tmp_path, release):
    with pytest.raises(FileNotFoundError):
        KEEPModel.from_pretrained(str(tmp_path))                               # no weights file there
    # the weights are read and held on the host until .to('cuda'); strict loading happens at upload
    model = AutoModel.from_pretrained(release)
    assert type(model) is KEEPModel and model._host_sd is not None and "visual.cls_token" in model._host_sd
    assert model.hf_config.model_type == "keep"

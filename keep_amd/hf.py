"""HuggingFace ``Auto*`` registration of the engine: the construction lines of the reference scripts run unchanged.

The reference registers its torch module under model type ``"keep"``
(``quick_start/keep_inference.py:75-76``) and builds it with ``AutoModel.from_config(config)`` (``:81``) or
``AutoModel.from_pretrained(model_path, trust_remote_code=True)`` (``WSI_evaluation/zeroshot_subtyping_WSI.py:44``).
Importing this module performs the same two registrations with :class:`keep_amd.KEEPModel` as the model class, so

    import keep_amd.hf                                    # instead of the KEEPConfig / KEEPModel class definitions
    config = AutoConfig.from_pretrained(model_path + 'config.json')
    model = AutoModel.from_config(config)                 # -> keep_amd.KEEPModel
    model.load_state_dict(torch.load(model_path + 'pytorch_model.bin', map_location='cpu'), strict=True)

and ``AutoModel.from_pretrained(local_release_dir).to(device)`` return the MI355X engine.  Only local directories can
be opened (``config.json`` must say ``"model_type": "keep"``; an ``auto_map`` entry pointing at remote code is ignored when
``trust_remote_code`` is left False -- with True, transformers would import the release's own torch module instead).
"""
from __future__ import annotations

from transformers import AutoConfig, AutoModel, PretrainedConfig

from .config import KEEPShape
from .model import DEFAULT_PRECISION, KEEPModel


class KEEPConfig(PretrainedConfig):
    """Same three fields as the reference config (``keep_inference.py:9-22``); ``vision_config`` is carried but unread there too
    (the ViT-L/16 constructor arguments are fixed in code, ``:32-40``)."""
    model_type = "keep"

    def __init__(self, vision_config=None, text_config=None, projection_dim=768, **kwargs):
        super().__init__(**kwargs)
        self.vision_config = vision_config
        self.text_config = text_config
        self.projection_dim = projection_dim

    def to_shape(self) -> KEEPShape:
        return KEEPShape.from_config_json({"text_config": self.text_config or {}, "projection_dim": self.projection_dim})


def _from_config(cls, config, **kwargs):
    """``AutoModel.from_config(config)`` -> an engine with no weights yet (the reference then calls ``load_state_dict``)."""
    kwargs.pop("trust_remote_code", None)
    kwargs.pop("torch_dtype", None)
    kwargs.pop("dtype", None)
    shape = config.to_shape() if isinstance(config, KEEPConfig) else KEEPShape.from_config_json(config.to_dict())
    model = cls(shape, precision=kwargs.pop("precision", DEFAULT_PRECISION), towers=kwargs.pop("towers", ("image", "text")))
    model.hf_config = config
    return model


_plain_from_pretrained = KEEPModel.from_pretrained.__func__


def _from_pretrained(cls, pretrained_model_name_or_path, *model_args, config=None, **kwargs):
    """``AutoModel.from_pretrained(dir, ...)`` hands over the parsed config and its hub keywords; only the directory matters here."""
    for k in ("trust_remote_code", "cache_dir", "force_download", "local_files_only", "token", "revision", "subfolder", "proxies",
              "code_revision", "adapter_kwargs", "torch_dtype", "dtype", "_from_auto", "_commit_hash", "device_map", "low_cpu_mem_usage",
              "use_safetensors", "weights_only", "attn_implementation"):
        kwargs.pop(k, None)
    model = _plain_from_pretrained(cls, str(pretrained_model_name_or_path), **kwargs)
    model.hf_config = config
    return model


KEEPModel.config_class = KEEPConfig
KEEPModel._from_config = classmethod(_from_config)
KEEPModel.from_pretrained = classmethod(_from_pretrained)

AutoConfig.register("keep", KEEPConfig, exist_ok=True)              # keep_inference.py:75
AutoModel.register(KEEPConfig, KEEPModel, exist_ok=True)            # keep_inference.py:76

__all__ = ["KEEPConfig", "KEEPModel"]

"""Host-side mirror of the reference's model API on top of libkeep_hip.

``KEEPModel`` keeps the surface of ``KEEPModel`` in ``quick_start/keep_inference.py:25-73`` (the
class HF ``AutoModel.from_pretrained('Astaxanthin/KEEP', trust_remote_code=True)`` returns in
``WSI_evaluation/zeroshot_*_WSI.py``): ``encode_image``, ``encode_text``, ``forward``, ``eval``,
``to``, ``load_state_dict``, ``logit_scale`` -- same argument meaning, same output shapes/dtypes,
L2-normalised fp32 features on the caller's device.  Underneath there is no torch module: tensors
are handed to the C ABI as raw device pointers; torch supplies storage, streams and file loading.

There is no CPU execution path.  Inputs that live on the CPU are copied to the engine's GPU and the
result is copied back; without a GPU (or without libkeep_hip.so) every call raises.
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os
import weakref
from typing import Dict, List, Mapping, Optional, Union

import torch

from . import _lib
from .config import KEEPShape

_PIX = {torch.float32: _lib.PIX_F32, torch.float16: _lib.PIX_F16, torch.bfloat16: _lib.PIX_BF16}
_PRECISIONS = {"fp16": _lib.PREC_FP16, "strict": _lib.PREC_STRICT, "comp": _lib.PREC_COMP}
DEFAULT_PRECISION = "comp"     # the mode that meets the reference tolerance (cosines within 1e-4) at the lowest cost
# 'comp' settings in order of cost: (comp_full_blocks, comp_mlp_blocks) = blocks whose attention side runs split products / whose MLP GEMMs
# carry the MX-fp4 correction terms.  calibrate() walks up this ladder until the probe's worst cosine error is inside its target.
COMP_LADDER = ((0, 0), (0, 4), (1, 4), (1, 6), (1, 8), (1, 10), (1, 12), (2, 12), (2, 16), (2, 24), (4, 24), (8, 24), (24, 24))
# calibrate() keeps the cheapest rung whose probe statistics predict a worst cosine error inside TOLERANCE over the POPULATION the model will
# be used on (tiles x distinct prompts): rms x expected_max_sigmas(population) <= TOLERANCE, and the probe's own maximum scaled the same way.
# Measured: config 3 (262 144 cosines) max / rms = 4.4-4.6 against expected_max_sigmas = 4.63; 100 000 tiles x 64 prompts: see bench.py c4.
CALIBRATION_POPULATION = 100_000 * 264      # BASELINE config 4: a 100 000-tile slide against the 264 distinct prompt strings of the RCC bank
TOLERANCE = 1e-4


def expected_max_sigmas(n: float) -> float:
    """E[max |x_i|] / sigma over n independent N(0, sigma) samples (extreme-value asymptote of 2n one-sided samples): the factor between the
    rms of the cosine errors and the worst one to expect in a population of n cosines.  16 384 -> 4.03, 262 144 -> 4.63, 6.4e6 -> 5.26,
    2.6e7 -> 5.51, 7.1e8 -> 6.06."""
    n2 = 2.0 * max(float(n), 2.0)
    a = math.sqrt(2.0 * math.log(n2))
    return a - (math.log(math.log(n2)) + math.log(4.0 * math.pi)) / (2.0 * a)


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream(device: torch.device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class KEEPModel:
    """Drop-in for the reference ``KEEPModel`` (inference only)."""

    # engines that own a handle, newest last: the slide-level functions of keep_amd.wsi take no model argument (the reference's are plain
    # torch code, WSI_evaluation/utils.py:107-146) and find the engine of their device here (keep_amd.model.engine_for)
    _live: "List[weakref.ref]" = []

    def __init__(self, config: Optional[Union[KEEPShape, Mapping, str]] = None, precision: str = DEFAULT_PRECISION,
                 towers=("image", "text")):
        if config is None:
            config = KEEPShape()
        elif not isinstance(config, KEEPShape):
            config = KEEPShape.from_config_json(config)
        self.config = config
        # nn.Parameter(torch.ones([]) * log(1/0.04)) -- keep_inference.py:52 (never applied at inference)
        self.logit_scale = torch.tensor(config.logit_scale_init, dtype=torch.float32)
        self.training = False
        self._handle = C.c_void_p(0)
        self._device: Optional[torch.device] = None
        self._host_sd: Optional[Dict[str, torch.Tensor]] = None
        self._loaded = False
        self._options = {"precision": _PRECISIONS[precision], "strict_blocks": 0}
        # load_state_dict(strict=True) demands the keys of every tower named here (the reference's model has both,
        # keep_inference.py:28-52); a single-tower engine -- e.g. an encode_image-only worker -- opts in with towers=("image",)
        self.towers = tuple(towers)
        if not self.towers or any(t not in ("image", "text") for t in self.towers):
            raise ValueError(f"towers must be a non-empty subset of ('image', 'text'), got {towers!r}")
        # out-of-range token ids raise IndexError, as nn.Embedding does.  True: checked before encode_text returns (one host
        # synchronisation per call); "lazy": the flag is copied back asynchronously and the error is raised by the NEXT engine call
        # or by check_errors() -- no synchronisation, which is also how the reference's CUDA path reports it (device-side assert);
        # False: never checked -- and that includes the non-finite-feature bit (an activation beyond the fp16 range of the engine's stores reaches the
        # output as NaN; with False nothing raises, the caller sees the NaNs).  The reference's WSI scripts make thousands of one-prompt calls, so the
        # default is "lazy".
        self.check_token_ids = "lazy"
        self._pending_token_checks = []   # [(pinned int32 flag, event)] in issue order; the device flag is sticky, so none can be lost
        self._flag_pool = []              # pinned buffers are recycled only after their copy has landed
        # load_state_dict on a GPU ends with calibrate(): the cheapest 'comp' setting whose worst cosine error on a seeded probe batch
        # (against the engine's own split-product arithmetic) predicts a worst error inside the tolerance over CALIBRATION_POPULATION cosines.  KEEP_CALIBRATE=0 / auto_calibrate=False: keep the
        # built-in default (1, 8), which was chosen on ONE synthetic weight family.
        self.auto_calibrate = os.environ.get("KEEP_CALIBRATE", "1") != "0"
        self.calibration: Optional[dict] = None
        self.trim_padding = True      # encode_text at the longest valid length instead of the padded one (same result)
        self.last_text_length = 0     # T the text tower actually ran at in the last encode_text call

    # ------------------------------------------------------------------ lifetime
    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    def _destroy(self):
        if getattr(self, "_handle", None) is not None and self._handle.value:
            _lib.load().keep_destroy(self._handle)
            self._handle = C.c_void_p(0)
            self._loaded = False

    def _create(self, device: torch.device):
        lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.KeepHipError("no GPU visible: keep_amd has no CPU execution path")
        if device.type != "cuda":
            raise ValueError(f"keep_amd runs on a ROCm GPU ('cuda:N'), not on {device}")
        idx = device.index if device.index is not None else torch.cuda.current_device()
        h = C.c_void_p(0)
        rc = lib.keep_create(idx, C.byref(h))
        if rc != _lib.KEEP_OK:
            raise _lib.KeepHipError(f"keep_create(device={idx}) failed with code {rc}")
        self._handle = h
        self._device = torch.device("cuda", idx)
        KEEPModel._live[:] = [r for r in KEEPModel._live if r() is not None and r() is not self] + [weakref.ref(self)]
        for k, v in self._options.items():
            _lib.check(h, lib.keep_set_option(h, k.encode(), float(v)), k)

    # ------------------------------------------------------------------ nn.Module-like surface
    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("keep_amd is an inference engine; training is out of scope")
        return self.eval()

    def requires_grad_(self, flag: bool = False):
        return self

    @property
    def device(self) -> torch.device:
        return self._device if self._device is not None else torch.device("cpu")

    def to(self, device=None, *args, **kwargs):
        if device is None:
            return self
        device = torch.device(device)
        if device.type == "cpu":
            if self._loaded:
                raise _lib.KeepHipError("keep_amd has no CPU execution path; the model stays on its GPU")
            return self
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
        if self._loaded and self._device == device:
            return self
        if self._loaded and self._host_sd is None:
            raise _lib.KeepHipError("weights were already uploaded and the host copy released; "
                                    "reload the state_dict to move to another GPU")
        self._destroy()
        self._create(device)
        if self._host_sd is not None:
            self._upload(self._host_sd)
            self._host_sd = None
        return self

    cuda = lambda self, device=None: self.to(torch.device("cuda", device) if isinstance(device, int) else (device or "cuda"))

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, state_dict: Mapping[str, torch.Tensor], strict: bool = True):
        """``model.load_state_dict(state_dict, strict=True)`` -- keep_inference.py:83.

        Key layout: SURVEY.md §A.3.  On a model that is already on a GPU the tensors are repacked
        immediately; otherwise they are held (by reference) until ``.to('cuda')``.
        """
        sd = {k: v for k, v in state_dict.items()}
        if "logit_scale" in sd:
            self.logit_scale = sd["logit_scale"].detach().to("cpu", torch.float32).reshape(())
        self._strict = strict
        if self._handle.value:
            self._upload(sd)
        else:
            self._host_sd = sd
        return self

    def _upload(self, sd: Mapping[str, torch.Tensor]):
        lib, h = _lib.load(), self._handle
        errors = []
        for key, t in sd.items():
            if not isinstance(t, torch.Tensor):
                continue
            t = t.detach()
            on_dev = t.device.type == "cuda" and t.device == self._device
            src = t.to(torch.float32).contiguous() if on_dev else t.to("cpu", torch.float32).contiguous()
            shape = (C.c_int64 * max(src.dim(), 1))(*src.shape)
            rc = lib.keep_load_tensor(h, key.encode(), _ptr(src), src.dim(), shape, 1 if on_dev else 0)
            if rc == _lib.KEEP_EKEY:
                if getattr(self, "_strict", True):
                    errors.append(lib.keep_last_error(h).decode())
            else:
                _lib.check(h, rc, key)
        if errors:
            raise RuntimeError("Error(s) in loading state_dict for KEEPModel:\n\t" + "\n\t".join(errors))
        notes = lib.keep_load_warnings(h).decode()
        if notes:
            import warnings
            for line in notes.split("\n"):
                warnings.warn(line, RuntimeWarning, stacklevel=3)
        rc = lib.keep_finalize_weights(h)
        if rc == _lib.KEEP_EKEY:
            raise RuntimeError("Error(s) in loading state_dict for KEEPModel:\n\t" + lib.keep_last_error(h).decode())
        _lib.check(h, rc, "finalize_weights")
        if getattr(self, "_strict", True):
            missing = []
            if "image" in self.towers and lib.keep_vit_depth(h) == 0:
                missing.append("visual.* / visual_head.* (image tower)")
            if "text" in self.towers and lib.keep_bert_layers(h) == 0:
                missing.append("text.* (text tower)")
            if missing:
                raise RuntimeError("Error(s) in loading state_dict for KEEPModel:\n\tMissing key(s) in state_dict: " + ", ".join(missing)
                                   + ' (construct with towers=("image",) / ("text",) for a single-tower engine)')
        self._loaded = True
        self._weights_epoch = getattr(self, "_weights_epoch", 0) + 1      # invalidates per-model prompt caches (keep_amd.wsi)
        self.calibration = None
        if self.auto_calibrate and self._options["precision"] == _lib.PREC_COMP and lib.keep_vit_depth(h) > 0:
            self.calibrate()

    @classmethod
    def from_pretrained(cls, path: str, precision: str = DEFAULT_PRECISION, **_ignored) -> "KEEPModel":
        """Load a release directory (``config.json`` + ``pytorch_model.bin`` or ``model.safetensors``),
        the local-files equivalent of ``AutoModel.from_pretrained`` at zeroshot_subtyping_WSI.py:44."""
        cfg_path = os.path.join(path, "config.json")
        config = KEEPShape.from_config_json(cfg_path) if os.path.exists(cfg_path) else KEEPShape()
        model = cls(config, precision=precision, towers=_ignored.pop("towers", ("image", "text")))
        st = os.path.join(path, "model.safetensors")
        pt = os.path.join(path, "pytorch_model.bin")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        elif os.path.exists(pt):
            sd = torch.load(pt, map_location="cpu")
        else:
            raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {path}")
        model.load_state_dict(sd, strict=True)
        return model

    # ------------------------------------------------------------------ options
    def set_precision(self, precision: str = DEFAULT_PRECISION, strict_blocks: int = 0):
        """'comp' (default): fp16 MFMA pass plus the first-order correction terms where the error budget needs them
        (MX-fp4 correction passes in the image tower's MLP GEMMs, split products in its first blocks and in the text
        tower): cosines within 1e-4 of the fp32 reference.  'fp16': single fp16 pass everywhere (fastest; ~1.5e-4, outside
        the tolerance -- opt-in).  'strict': hi/lo split operands, three MFMA passes (4e-7, ~0.4x the speed).
        ``strict_blocks=n`` additionally runs the first n transformer blocks of each tower in split mode."""
        self._options["precision"] = _PRECISIONS[precision]
        self._options["strict_blocks"] = int(strict_blocks)
        if self._handle.value:
            lib = _lib.load()
            for k, v in self._options.items():
                _lib.check(self._handle, lib.keep_set_option(self._handle, k.encode(), float(v)), k)
        return self

    def set_option(self, name: str, value: float):
        self._options[name] = value
        if self._handle.value:
            _lib.check(self._handle, _lib.load().keep_set_option(self._handle, name.encode(), float(value)), name)
        return self

    @torch.no_grad()
    def calibrate(self, n_tiles: int = 256, population: Optional[float] = None, tiles: Optional[torch.Tensor] = None,
                  text_features: Optional[torch.Tensor] = None, seed: int = 20250929, tolerance: float = TOLERANCE) -> Optional[dict]:
        """Pick the 'comp' setting for THESE weights and for the population the model will be used on.

        A probe batch (``tiles``, default ``n_tiles`` seeded N(0,1) tiles -- what ImageNet-normalised pixels look like) is encoded
        once with split products (the engine's fp32-class arithmetic, ~5e-7 from the fp32 reference) and then with each rung of
        ``COMP_LADDER`` from the cheapest up.  Kept: the first rung whose cosine errors over probe tiles x prompts predict a worst
        error <= ``tolerance`` (1e-4) over ``population`` cosines (tiles x distinct prompts the caller will compare; default
        ``CALIBRATION_POPULATION`` = a 100 000-tile slide x the 264 distinct prompts of the RCC bank, BASELINE config 4):
        ``rms x expected_max_sigmas(population) <= tolerance`` and ``probe max x sigmas(population) / sigmas(probe size) <= tolerance``.
        The prompts are ``text_features`` ([P,768] unit rows -- pass the caller's own bank to calibrate against it; default: 64 seeded
        prompts through the loaded text tower, or 64 seeded random unit vectors for an image-only engine, a harsher yardstick).  If even
        the last rung misses, the engine switches to 'strict'.  Non-finite probe features (an activation beyond the fp16 range) raise
        FloatingPointError.  The precision options the model had (``strict_blocks`` included) are kept; on an exception the previous
        setting is restored.  Returns and stores ``self.calibration``."""
        lib, h = _lib.load(), self._handle
        if not self._loaded and self._host_sd is not None:
            self.to("cuda")
            if self.calibration is not None:
                return self.calibration
        if not h.value or lib.keep_vit_depth(h) == 0 or self._options["precision"] != _lib.PREC_COMP:
            return None
        dev, depth = self._device, int(lib.keep_vit_depth(h))
        population = float(CALIBRATION_POPULATION if population is None else population)
        if tiles is None:
            g = torch.Generator(device=dev).manual_seed(seed)
            tiles = torch.randn(n_tiles, 3, 224, 224, device=dev, generator=g, dtype=torch.float32).to(torch.bfloat16)
        tiles = tiles.to(dev)
        if text_features is None:
            if lib.keep_bert_layers(h) > 0:
                from .synth import synth_prompts
                toks = synth_prompts(64, 64, seed=seed % 100003, vocab=self.config.text.vocab_size)
                text_features = self.encode_text({k: v.to(dev) for k, v in toks.items()})
            else:
                g = torch.Generator().manual_seed(seed + 1)
                text_features = torch.randn(64, self.config.projection_dim, generator=g).to(dev)
                _lib.check(h, lib.keep_op_l2norm(h, _ptr(text_features), 64, self.config.projection_dim, _stream(dev)), "l2norm")
        bank = text_features.to(dev, torch.float32).contiguous()
        n_probe = tiles.shape[0] * bank.shape[0]
        z_pop, z_probe = expected_max_sigmas(population), expected_max_sigmas(n_probe)
        rms_target, max_target = tolerance / z_pop, tolerance * min(1.0, z_probe / z_pop)
        saved = {k: self._options.get(k) for k in ("precision", "strict_blocks", "comp_full_blocks", "comp_mlp_blocks")}
        strict_blocks = int(saved["strict_blocks"] or 0)
        was, self.auto_calibrate = self.auto_calibrate, False
        done = False
        try:
            self.set_precision("strict", strict_blocks)
            ref = self.similarity(self.encode_image(tiles), bank)           # cosines on the engine's exact-fp32 similarity kernel
            self.set_precision("comp", strict_blocks)
            if not bool(torch.isfinite(ref).all()):
                self._raise_flags(2)
            tried, chosen = [], None
            rungs = list(dict.fromkeys((min(a, depth), min(b, depth)) for a, b in COMP_LADDER))
            for full, mlp in rungs:
                self.set_option("comp_full_blocks", full)
                self.set_option("comp_mlp_blocks", mlp)
                d = self.similarity(self.encode_image(tiles), bank).sub_(ref).abs_()
                err, rms = float(d.max()), float(d.pow(2).mean().sqrt())
                tried.append({"comp_full_blocks": full, "comp_mlp_blocks": mlp, "max_abs_dcos": float(f"{err:.3e}"), "rms_dcos": float(f"{rms:.3e}")})
                if err <= max_target and rms <= rms_target:      # (NaN compares False: falls through to the next rung)
                    chosen = (full, mlp)
                    break
            if chosen is None:
                self.set_precision("strict", strict_blocks)
            self.check_errors(wait=True)
            done = True
        finally:
            self.auto_calibrate = was
            if not done:                 # an exception mid-ladder: put back exactly what the caller had
                for k, v in saved.items():
                    if v is None:
                        self._options.pop(k, None)
                    else:
                        self._options[k] = v
                if self._handle.value:
                    for k, v in self._options.items():
                        lib.keep_set_option(self._handle, k.encode(), float(v))
        self.calibration = {"precision": "comp" if chosen else "strict", "comp_full_blocks": chosen[0] if chosen else None,
                            "comp_mlp_blocks": chosen[1] if chosen else None, "population": population,
                            "expected_max_sigmas": round(z_pop, 3), "target_max_abs_dcos": float(f"{max_target:.3e}"),
                            "target_rms_dcos": float(f"{rms_target:.3e}"), "strict_blocks": strict_blocks,
                            "probe": f"{tiles.shape[0]} tiles x {bank.shape[0]} prompts vs the split-product arithmetic", "tried": tried}
        return self.calibration

    def get_option(self, name: str) -> float:
        self._ready_device()
        return float(_lib.load().keep_get_option(self._handle, name.encode()))

    def reserve(self, tiles: int = 0, prompts: int = 0, seq: int = 256):
        self._ready()
        _lib.check(self._handle, _lib.load().keep_reserve(self._handle, tiles, prompts, seq), "reserve")
        return self

    def _ready(self):
        self.check_errors(wait=False)
        if not self._loaded:
            if self._host_sd is not None:
                self.to("cuda")            # reference scripts call .to(device) themselves; be lenient
            else:
                raise _lib.KeepHipError("no weights loaded: call load_state_dict / from_pretrained first")

    # ------------------------------------------------------------------ the hot path
    @torch.no_grad()
    def encode_image(self, image_inputs: torch.Tensor) -> torch.Tensor:
        """keep_inference.py:54-58: normalize(visual_head(visual(x)), dim=-1) -> [B, 768] fp32."""
        self._ready()
        x = image_inputs
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"expected [B,3,224,224], got {tuple(x.shape)}")
        if x.shape[2] != 224 or x.shape[3] != 224:
            raise ValueError("only 224x224 tiles are supported (the reference would resample pos_embed via "
                             "timm dynamic_img_size; every reference caller feeds 224x224)")
        if x.dtype not in _PIX:
            x = x.to(torch.float32)
        src_dev = x.device
        if x.shape[0] == 0:
            return torch.empty((0, self.config.projection_dim), dtype=torch.float32, device=src_dev)
        xd = x.to(self._device, non_blocking=True).contiguous()
        out = torch.empty((xd.shape[0], self.config.projection_dim), dtype=torch.float32, device=self._device)
        lib = _lib.load()
        _lib.check(self._handle, lib.keep_encode_image(self._handle, _ptr(xd), _PIX[xd.dtype], xd.shape[0], _ptr(out),
                                                       _stream(self._device)), "encode_image")
        self._queue_flag_check(_stream(self._device))
        if src_dev == self._device:
            return out
        res = out.to(src_dev)
        self.check_errors(wait=True)
        return res

    @torch.no_grad()
    def encode_image_uint8(self, tiles_u8: torch.Tensor) -> torch.Tensor:
        """Raw RGB tiles, uint8 [B,224,224,3] (HWC, after the resize + centre crop of the reference transform):
        ToTensor and Normalize (keep_inference.py:91-92) are fused into the first kernel, so the host only ships
        150 KB per tile.  Same result as ``encode_image(normalised_float_tiles)`` up to fp32 rounding."""
        self._ready()
        x = tiles_u8
        if x.dtype != torch.uint8 or x.dim() != 4 or tuple(x.shape[1:]) != (224, 224, 3):
            raise ValueError(f"expected uint8 [B,224,224,3], got {x.dtype} {tuple(x.shape)}")
        src_dev = x.device
        if x.shape[0] == 0:
            return torch.empty((0, self.config.projection_dim), dtype=torch.float32, device=src_dev)
        xd = x.to(self._device, non_blocking=True).contiguous()
        out = torch.empty((xd.shape[0], self.config.projection_dim), dtype=torch.float32, device=self._device)
        _lib.check(self._handle, _lib.load().keep_encode_image(self._handle, _ptr(xd), _lib.PIX_U8_HWC, xd.shape[0], _ptr(out),
                                                               _stream(self._device)), "encode_image_uint8")
        self._queue_flag_check(_stream(self._device))
        if src_dev == self._device:
            return out
        res = out.to(src_dev)                  # host inputs: the copy back synchronises anyway, so the check is free and immediate
        self.check_errors(wait=True)
        return res

    @torch.no_grad()
    def resize_crop_uint8(self, images_u8: torch.Tensor, size: int = 224) -> torch.Tensor:
        """Raw RGB images, uint8 [B,H,W,3] of ONE size -> uint8 [B,size,size,3]: the reference transform's
        ``Resize(size, BICUBIC)`` + ``CenterCrop((size, size))`` (keep_inference.py:88-90) on the device, bit-identical to the
        PIL code path torchvision runs.  Feed the result to :meth:`encode_image_uint8`."""
        from .preprocess import pil_bicubic_coeffs, resize_output_size
        self._ready_device()
        x = images_u8
        if x.dtype != torch.uint8 or x.dim() != 4 or x.shape[3] != 3:
            raise ValueError(f"expected uint8 [B,H,W,3], got {x.dtype} {tuple(x.shape)}")
        B, H, W, _ = x.shape
        ow, oh = resize_output_size(W, H, size)
        if ow < size or oh < size:
            raise ValueError(f"resized image {ow}x{oh} is smaller than the {size}-pixel crop (torchvision would zero-pad: not supported on the device)")
        key = (W, H, size)
        if getattr(self, "_resize_cache", None) is None:
            self._resize_cache = {}
        if key not in self._resize_cache:             # float64 table construction as in libImaging; a few ms, once per image size
            xb, xk, xks = pil_bicubic_coeffs(W, ow)
            yb, yk, yks = pil_bicubic_coeffs(H, oh)
            dev = lambda a: torch.from_numpy(a).to(self._device).contiguous()
            self._resize_cache[key] = (dev(xb), dev(xk), xks, dev(yb), dev(yk), yks)
        xb, xk, xks, yb, yk, yks = self._resize_cache[key]
        left, top = int(round((ow - size) / 2.0)), int(round((oh - size) / 2.0))
        src_dev = x.device
        xd = x.to(self._device, non_blocking=True).contiguous()
        out = torch.empty((B, size, size, 3), dtype=torch.uint8, device=self._device)
        _lib.check(self._handle, _lib.load().keep_resize_crop_u8(self._handle, _ptr(xd), B, H, W, _ptr(xb), _ptr(xk), xks, ow, _ptr(yb), _ptr(yk), yks, oh,
                                                                 left, top, size, _ptr(out), _stream(self._device)), "resize_crop_u8")
        return out if src_dev == self._device else out.to(src_dev)

    @torch.no_grad()
    def encode_image_raw(self, images_u8: torch.Tensor) -> torch.Tensor:
        """uint8 [B,H,W,3] raw tiles of any (common) size -> [B,768]: the whole reference transform + ``encode_image`` on the
        device (resize + crop bit-identical to PIL, /255 and mean/std fused into the first encoder kernel)."""
        self._ready()
        dev_in = images_u8.device
        out = self.encode_image_uint8(self.resize_crop_uint8(images_u8.to(self._device)))
        return out if dev_in == self._device else out.to(dev_in)

    @torch.no_grad()
    def encode_text(self, text_inputs: Mapping[str, torch.Tensor]) -> torch.Tensor:
        """keep_inference.py:60-62: normalize(text(**inputs).pooler_output, dim=-1) -> [P, 768] fp32."""
        self._ready()
        ids = text_inputs["input_ids"]
        if ids.dim() != 2:
            raise ValueError(f"input_ids must be [P,T], got {tuple(ids.shape)}")
        src_dev = ids.device
        if ids.shape[0] == 0:
            return torch.empty((0, self.config.text.hidden_size), dtype=torch.float32, device=src_dev)

        def prep(name):
            t = text_inputs.get(name) if hasattr(text_inputs, "get") else (text_inputs[name] if name in text_inputs else None)
            if t is None:
                return None
            if tuple(t.shape) != tuple(ids.shape):
                raise ValueError(f"{name} shape {tuple(t.shape)} != input_ids shape {tuple(ids.shape)}")
            return t.to(self._device, torch.int64, non_blocking=True).contiguous()

        ids_d = ids.to(self._device, torch.int64, non_blocking=True).contiguous()
        typ_d, msk_d = prep("token_type_ids"), prep("attention_mask")
        P, T = ids_d.shape
        self.last_text_length = T
        if self.trim_padding and msk_d is not None and T > 16:
            # Columns that are padding in EVERY row change nothing (their softmax weight is exactly 0, positions are
            # absolute, the pooler reads token 0): run the tower at the longest valid length, rounded up to 16.
            # A row with no valid token attends uniformly over all T keys in HF, so such a batch is left alone.
            # The length is taken from the caller's mask where it lives: a host mask (the tokenizer's output) costs no device sync.
            src_mask = text_inputs.get("attention_mask") if hasattr(text_inputs, "get") else text_inputs["attention_mask"]
            valid = (src_mask if src_mask.device.type == "cpu" else msk_d) != 0
            col = torch.nonzero(valid.any(dim=0)).flatten()
            info = torch.stack([col[-1] + 1 if col.numel() else torch.zeros((), dtype=torch.int64, device=valid.device),
                                (~valid.any(dim=1)).any().to(torch.int64)]).tolist()
            L = min(T, max(16, -(-info[0] // 16) * 16))
            if info[1] == 0 and L < T:
                ids_d, msk_d = ids_d[:, :L].contiguous(), msk_d[:, :L].contiguous()
                typ_d = typ_d[:, :L].contiguous() if typ_d is not None else None
                T = self.last_text_length = L
        out = torch.empty((P, self.config.text.hidden_size), dtype=torch.float32, device=self._device)
        lib = _lib.load()
        st = _stream(self._device)
        _lib.check(self._handle, lib.keep_encode_text(self._handle, _ptr(ids_d), _ptr(typ_d), _ptr(msk_d), P, T,
                                                      _ptr(out), st), "encode_text")
        self._queue_flag_check(st)
        if src_dev == self._device:
            return out
        res = out.to(src_dev)                  # host inputs: the copy back synchronises anyway, so the check is free and immediate
        self.check_errors(wait=True)
        return res

    def _queue_flag_check(self, st):
        """After an encode call: look at the handle's sticky error bits (out-of-range token ids, non-finite features) -- immediately
        (``check_token_ids=True``), not at all (False), or lazily: an asynchronous copy of the bits + an event, examined by the next
        engine call / ``check_errors()`` (no host synchronisation per call)."""
        lib = _lib.load()
        if self.check_token_ids == "lazy":
            if len(self._pending_token_checks) >= 64:                  # bound the queue: wait for the oldest copies
                self.check_errors(wait=True)
            flag = self._flag_pool.pop() if self._flag_pool else torch.zeros(1, dtype=torch.int32).pin_memory()
            _lib.check(self._handle, lib.keep_token_error_async(self._handle, _ptr(flag), st), "token_error_async")
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self._device))
            self._pending_token_checks.append((flag, ev))
        elif self.check_token_ids:
            self._raise_flags(self._error_bits(st))

    def _error_bits(self, st) -> int:
        """keep_token_error: the sticky error bits (>= 0, cleared by the call) or a negative error code -- a HIP failure must not be read as bits."""
        rc = int(_lib.load().keep_token_error(self._handle, st))
        if rc < 0:
            _lib.check(self._handle, rc, "token_error")
        return rc

    @staticmethod
    def _raise_flags(bits: int, earlier: bool = False):
        when = " of an earlier call" if earlier else ""
        if bits & 1:
            raise IndexError(f"index out of range in self (input_ids / token_type_ids{when} were outside the embedding tables)")
        if bits & 2:
            raise FloatingPointError(f"non-finite output features{when}: an activation exceeded the fp16 range (65504) of the engine's qkv / "
                                     "MLP-hidden stores; these weights need the fp32 reference path")

    def check_errors(self, wait: bool = True):
        """Raise the error of an earlier encode call (lazy checking): IndexError for out-of-range token ids, FloatingPointError for
        non-finite features.  Called by every engine entry point with ``wait=False`` (only looks at copies that have already landed).
        The device-side bits are sticky and every call's copy is queued, so an error is reported by the first check after it --
        never dropped."""
        pend = self._pending_token_checks
        if not pend:
            return
        if wait:
            pend[-1][1].synchronize()
        bits = 0
        while pend and pend[0][1].query():
            flag, _ = pend.pop(0)
            bits |= int(flag.item())
            self._flag_pool.append(flag)
        if bits:
            # acknowledge (clears the sticky bits); copies still in flight were taken before the clear and would repeat the report
            bits |= self._error_bits(_stream(self._device))
            for flag, ev in pend:
                ev.synchronize()
                self._flag_pool.append(flag)
            pend.clear()
            self._raise_flags(bits, earlier=True)

    def forward(self, image_inputs, text_inputs):
        """keep_inference.py:65-73."""
        return {"vision_features": self.encode_image(image_inputs),
                "text_features": self.encode_text(text_inputs)}

    __call__ = forward

    # ------------------------------------------------------------------ similarity
    @torch.no_grad()
    def similarity(self, image_features: torch.Tensor, text_features: torch.Tensor, scale: float = 1.0,
                   mode: str = "raw"):
        """Tile x prompt similarity on the GPU.

        mode 'raw'     -> scale * I @ T^T                       [N,P] fp32   (keep_inference.py:104)
             'argmax'  -> (sim [N,P], labels [N] int32)
             'softmax' -> softmax(scale * I @ T^T, dim=1)       [N,P] fp32   (subtyping_utils.py:72, scale=10)
             'softmax_f16' -> same in fp16
             'top2score'   -> python float, rank_cls_score of I @ T^T (WSI_evaluation/utils.py:107-117)
        """
        self._ready_device()
        img = image_features.to(self._device, torch.float32).contiguous()
        txt = text_features.to(self._device, torch.float32).contiguous()
        if img.dim() != 2 or txt.dim() != 2 or img.shape[1] != txt.shape[1]:
            raise ValueError(f"feature shapes {tuple(img.shape)} x {tuple(txt.shape)}")
        N, D = img.shape
        P = txt.shape[0]
        lib, st = _lib.load(), _stream(self._device)
        code = {"raw": _lib.SIM_RAW, "argmax": _lib.SIM_ARGMAX, "softmax": _lib.SIM_SOFTMAX,
                "softmax_f16": _lib.SIM_SOFTMAX_F16, "top2score": _lib.SIM_TOP2SCORE}[mode]
        amax = None
        if mode == "top2score":
            out = torch.empty((1,), dtype=torch.float32, device=self._device)
        elif mode == "softmax_f16":
            out = torch.empty((N, P), dtype=torch.float16, device=self._device)
        else:
            out = torch.empty((N, P), dtype=torch.float32, device=self._device)
        if mode == "argmax":
            amax = torch.empty((N,), dtype=torch.int32, device=self._device)
        _lib.check(self._handle, lib.keep_similarity(self._handle, _ptr(img), _ptr(txt), N, P, D, float(scale), code,
                                                     _ptr(out), _ptr(amax), st), "similarity")
        if mode == "argmax":
            return out, amax
        if mode == "top2score":
            return float(out.item())
        return out

    @torch.no_grad()
    def classify(self, image_inputs: torch.Tensor, text_features: torch.Tensor, scale: float = 1.0, margin: Optional[float] = None,
                 return_features: bool = False):
        """Tiles -> (similarity [B,P] fp32, labels [B] int32[, features [B,768]]): ``encode_image`` + ``img @ txt.T`` + row argmax
        (keep_inference.py:101-104) with labels that are the fp32 reference's.  Every tile is encoded in the model's precision; the
        tiles whose two best prompts are closer than ``margin`` in cosine (default: the engine's ``label_margin``, 2.5e-4 = twice the
        1e-4 tolerance + 25 %) are encoded a second time with split products (``strict``) and take their row from that.
        ``self.last_rechecked`` holds how many tiles that was.  ``text_features``: ``encode_text`` output ([P,768], unit norm)."""
        self._ready()
        x = image_inputs
        u8 = x.dtype == torch.uint8
        if u8:
            if x.dim() != 4 or tuple(x.shape[1:]) != (224, 224, 3):
                raise ValueError(f"uint8 tiles must be [B,224,224,3] (HWC), got {tuple(x.shape)}")
        else:
            if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != 224 or x.shape[3] != 224:
                raise ValueError(f"expected [B,3,224,224], got {tuple(x.shape)}")
            if x.dtype not in _PIX:
                x = x.to(torch.float32)
        src_dev = x.device
        txt = text_features.to(self._device, torch.float32).contiguous()
        D = self.config.projection_dim
        if txt.dim() != 2 or txt.shape[1] != D:
            raise ValueError(f"text_features must be [P,{D}], got {tuple(txt.shape)}")
        B, P = x.shape[0], txt.shape[0]
        sim = torch.empty((B, P), dtype=torch.float32, device=self._device)
        lab = torch.empty((B,), dtype=torch.int32, device=self._device)
        feats = torch.empty((B, D), dtype=torch.float32, device=self._device) if return_features else None
        n = C.c_int64(0)
        if B:
            xd = x.to(self._device, non_blocking=True).contiguous()
            _lib.check(self._handle, _lib.load().keep_classify(self._handle, _ptr(xd), _lib.PIX_U8_HWC if u8 else _PIX[xd.dtype], B, _ptr(txt), P,
                                                               float(scale), -1.0 if margin is None else float(margin), _ptr(feats), _ptr(sim),
                                                               _ptr(lab), C.byref(n), _stream(self._device)), "classify")
            self._queue_flag_check(_stream(self._device))
        self.last_rechecked = int(n.value)
        out = (sim, lab) + ((feats,) if return_features else ())
        if src_dev == self._device:
            return out
        out = tuple(t.to(src_dev) for t in out)
        self.check_errors(wait=True)
        return out

    def _ready_device(self):
        self.check_errors(wait=False)
        if not self._handle.value:
            self._create(torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0))

    def clock_probe(self, out: torch.Tensor, spin_us: int = 300, stream: Optional["torch.cuda.Stream"] = None) -> None:
        """Enqueue a shader-clock sample on ``stream`` (default: current) into ``out`` (device int64[2], allocated -- and its allocation
        synchronised -- BEFORE the work the probe runs beside was queued: a fresh tensor handed to a side stream can be a recycled block
        that kernels still pending on the allocating stream write to).  out = {shader cycles, 100 MHz ticks}; MHz = 100 * cycles / ticks."""
        self._ready_device()
        if out.dtype != torch.int64 or out.numel() < 2 or out.device != self._device or not out.is_contiguous():
            raise ValueError("clock_probe needs a contiguous int64[2] on the engine's device")
        st = C.c_void_p(stream.cuda_stream) if stream is not None else _stream(self._device)
        _lib.check(self._handle, _lib.load().keep_clock_probe(self._handle, int(spin_us), _ptr(out), st), "clock_probe")

    def mfma_ceiling(self, operands: Optional[torch.Tensor] = None, iters: int = 100_000, reps: int = 3) -> float:
        """TFLOP/s the matrix pipes alone sustain on this GPU with ``operands`` (fp16 values; default seeded N(0, 1)) and no memory traffic
        (``keep_mfma_probe``): the ceiling the socket's power cap leaves for high-entropy fp16 MFMA work -- context for roofline fractions that
        are quoted against the nominal 2.4 GHz peak."""
        self._ready_device()
        dev = self._device
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        n = cus * 512 * 48
        if operands is None:
            g = torch.Generator(device=dev).manual_seed(7)
            operands = torch.randn(n, device=dev, generator=g, dtype=torch.float32).to(torch.float16)
        src = operands.to(dev, torch.float16).contiguous().flatten()
        if src.numel() < n:
            src = src.repeat(-(-n // src.numel()))[:n].contiguous()
        sink = torch.empty(cus * 512, dtype=torch.float32, device=dev)
        fl = C.c_double(0)
        best = 0.0
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(torch.cuda.current_stream(dev))
            _lib.check(self._handle, _lib.load().keep_mfma_probe(self._handle, _ptr(src), _ptr(sink), int(iters), C.byref(fl), _stream(dev)), "mfma_probe")
            b.record(torch.cuda.current_stream(dev))
            b.synchronize()
            best = max(best, fl.value / (a.elapsed_time(b) * 1e-3) / 1e12)
        return best

    # ------------------------------------------------------------------ profiling passthrough
    def profile_enable(self, tag: Optional[str] = None):
        self._ready_device()
        arg = None if tag is None else tag.encode()
        _lib.check(self._handle, _lib.load().keep_profile_enable(self._handle, arg), "profile_enable")

    def profile_disable(self):
        self.profile_enable("")

    def profile_reset(self):
        _lib.check(self._handle, _lib.load().keep_profile_reset(self._handle), "profile_reset")

    def profile_read(self, tag: str):
        ms, n, fl = C.c_double(0), C.c_int64(0), C.c_double(0)
        _lib.check(self._handle, _lib.load().keep_profile_read(self._handle, tag.encode(), C.byref(ms), C.byref(n), C.byref(fl)), tag)
        return ms.value, n.value, fl.value


_bare_engines: Dict[int, "KEEPModel"] = {}


def engine_for(device=None, model=None) -> "KEEPModel":
    """The engine the model-free slide-level functions run on: ``model`` if given (a KEEPModel or the reference's ``KEEP_model``
    dict), else the most recently created live engine on ``device`` (default: the current GPU), else a weight-less handle on
    that device (similarity / screening / refine kernels need no weights)."""
    if model is not None:
        m = model["model"] if isinstance(model, Mapping) else model
        if not isinstance(m, KEEPModel):
            raise TypeError("expected a keep_amd.KEEPModel (or the reference's KEEP_model dict holding one)")
        m._ready_device()
        return m
    if not torch.cuda.is_available():
        raise _lib.KeepHipError("no GPU visible: keep_amd has no CPU execution path")
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    if dev.type != "cuda":
        dev = torch.device("cuda", torch.cuda.current_device())      # features on the host: computed on the current GPU, returned to the host
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    for r in reversed(KEEPModel._live):
        m = r()
        if m is not None and m._handle.value and m._device is not None and m._device.index == idx:
            return m
    m = KEEPModel()
    m._create(torch.device("cuda", idx))
    _bare_engines[idx] = m                    # keeps the weight-less handle alive (the registry only holds weak references)
    return m


PROFILE_TAGS = ("vit.im2col", "vit.patch", "vit.ln", "vit.qkv", "vit.attn", "vit.proj", "vit.fc1", "vit.fc2",
                "vit.head", "text.embed", "text.ln", "text.qkv", "text.attn", "text.out", "text.ffn1", "text.ffn2",
                "text.pool", "sim",
                # launches with extra passes (split / compensated products of the blocks the precision setting names) and the CLS-rows-only
                # operators of the last block are timed apart, so that the plain tags hold ONE kernel instantiation each
                "vit.qkv.x", "vit.attn.x", "vit.proj.x", "vit.fc1.x", "vit.fc2.x", "vit.tail")

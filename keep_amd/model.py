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
from typing import Dict, List, Mapping, Optional, Sequence, Tuple, Union

import torch

from . import _lib
from .config import KEEPShape

_PIX = {torch.float32: _lib.PIX_F32, torch.float16: _lib.PIX_F16, torch.bfloat16: _lib.PIX_BF16}
_PRECISIONS = {"fp16": _lib.PREC_FP16, "strict": _lib.PREC_STRICT, "comp": _lib.PREC_COMP}
DEFAULT_PRECISION = "comp"     # the mode that meets the reference tolerance (cosines within 1e-4) at the lowest cost
# 'comp' spends its precision budget block by block (keep_set_block_precision): a PLAN is one (attention-side mode, MLP mode) pair per ViT block,
# modes as in include/keep_hip.h (KEEP_ATTN_*, KEEP_MLP_*).  The prefix shorthand (comp_full_blocks, comp_mlp_blocks) = the first blocks' attention
# side as split products / the first blocks' MLP GEMMs with both MX-fp4 correction terms is one family of plans; calibrate() walks it from the
# cheapest rung up (budget="ladder") or builds a plan from variance shares measured on the loaded weights (budget="measured").
COMP_LADDER = ((0, 0), (0, 4), (1, 4), (1, 6), (1, 8), (2, 8), (1, 10), (2, 10), (1, 12), (2, 12), (2, 14), (2, 16), (2, 24), (4, 24), (8, 24), (24, 24))
# calibrate() keeps the cheapest plan whose probe statistics predict, with probability CONFIDENCE, a worst cosine error inside TOLERANCE over the
# POPULATION the model will be used on (tiles x distinct prompts): rms x max_sigmas_quantile(population, CONFIDENCE) x tail factor <= TOLERANCE.
CALIBRATION_POPULATION = 100_000 * 264      # BASELINE config 4: a 100 000-tile slide against the 264 distinct prompt strings of the RCC bank
TOLERANCE = 1e-4
CONFIDENCE = 0.99
# The probe's rms is itself an estimate: 256 tiles x 64 prompts are 16 384 cosines, >= 4 096 effective samples once the correlation of one tile's
# cosines is allowed for, i.e. a relative standard error <= 1.1 %.  The rule holds the tolerance against rms x this factor (two standard errors).
PROBE_RMS_MARGIN = 1.02
# Milliseconds a knob adds to a 256-tile encode step on an MI355X (two lanes; tools/precision_budget.py measures them, profiles/r05_precision_budget.md):
# what the greedy of budget="measured" divides a knob's variance share by.  Only the RATIOS matter.
KNOB_COST_MS = {"attn_split": 0.92, "attn_split_compqkv": 0.56, "attn_compqkv": 0.22, "attn_proj_cls": 0.012, "attn_compqkv_proj_cls": 0.18, "mlp_comp": 0.52, "mlp_comp_w": 0.29, "mlp_cls": 0.07}
# Share of a site's rounding variance that survives a treatment when it is not measured on the loaded weights (same tool, bench weights): both
# MX-fp4 correction terms remove ~96 % of an MLP's share, the W_lo term alone 40-55 %; a compensated qkv inside a split attention side leaves 1-2 %
MLP_RESIDUAL = {_lib.MLP_PLAIN: 1.0, _lib.MLP_CLS: 0.1, _lib.MLP_COMP_W: 0.55, _lib.MLP_COMP: 0.04, _lib.MLP_SPLIT: 0.0}
ATTN_RESIDUAL = {_lib.ATTN_PLAIN: 1.0, _lib.ATTN_COMPQKV: 0.6, _lib.ATTN_PROJ_CLS: 0.55, _lib.ATTN_COMPQKV_PROJ_CLS: 0.25, _lib.ATTN_SPLIT_COMPQKV: 0.015, _lib.ATTN_SPLIT: 0.0}
# The treatments the greedy of budget="measured" may use.  Measured on the bench weights (profiles/r05_precision_budget.md): the W_lo-only MLP form
# removes ~45 % of a block's MLP share for 57 % of the cost of both terms, and a compensated qkv alone ~35 % of an attention side's share at 1 ms per
# percent of variance against 0.45 for MLP blocks -- neither ever wins a greedy step, so they are off by default (``knobs=`` switches them on).
# KEEP_MLP_CLS (every row plain, the CLS row of every tile again as split products) is the cheap one: the feature is pooled from the CLS rows, whose
# own rounding errors reach it directly while the other 196 rows' only arrive through attention averages -- what it leaves of a block's MLP share is
# measured per block (0.5-2 % on the bench weights).
# KEEP_ATTN_PROJ_CLS (round 6) is the attention side's counterpart: on spatially correlated tiles ~70 % of an attention side's rounding error is its proj
# GEMM's (tools/attn_site_study.py), and redoing the CLS row's proj as a split product on its fp32-grade attention output removes what a full CLS-row
# treatment of the attention side would -- for one more small GEMM in the CLS-row chain (measured per block and tile group, like KEEP_MLP_CLS).
DEFAULT_KNOBS = {"attn": (_lib.ATTN_PROJ_CLS, _lib.ATTN_COMPQKV_PROJ_CLS, _lib.ATTN_SPLIT_COMPQKV, _lib.ATTN_SPLIT), "mlp": (_lib.MLP_CLS, _lib.MLP_COMP)}

Plan = List[Tuple[int, int]]


def prefix_plan(depth: int, full_blocks: int, mlp_blocks: int) -> Plan:
    """The plan the (comp_full_blocks, comp_mlp_blocks) shorthand stands for."""
    return [(_lib.ATTN_SPLIT if i < full_blocks else _lib.ATTN_PLAIN, _lib.MLP_COMP if i < mlp_blocks else _lib.MLP_PLAIN) for i in range(depth)]


def plan_string(plan: Plan) -> str:
    """'attn:1000... mlp:2222...' -- one digit per block (KEEP_ATTN_* / KEEP_MLP_*)."""
    return "attn:" + "".join(str(a) for a, _ in plan) + " mlp:" + "".join(str(m) for _, m in plan)


def plan_prefix(plan: Plan) -> Optional[Tuple[int, int]]:
    """(comp_full_blocks, comp_mlp_blocks) if the plan is one of the prefix family, else None."""
    full = sum(1 for a, _ in plan if a == _lib.ATTN_SPLIT)
    mlp = sum(1 for _, m in plan if m == _lib.MLP_COMP)
    return (full, mlp) if plan == prefix_plan(len(plan), full, mlp) else None


def max_sigmas_gumbel(n: float) -> Tuple[float, float]:
    """(a, b) of the extreme-value (Gumbel) law of M = max_i |x_i| / sigma over n independent N(0, sigma) samples (2n one-sided samples):
    P(M <= z) ~ exp(-exp(-a (z - b))).  b is the LOCATION (the mode; M exceeds it 63 % of the time), the mean is b + 0.5772 / a."""
    n2 = 2.0 * max(float(n), 2.0)
    a = math.sqrt(2.0 * math.log(n2))
    return a, a - (math.log(math.log(n2)) + math.log(4.0 * math.pi)) / (2.0 * a)


def expected_max_sigmas(n: float) -> float:
    """LOCATION b_n of the maximum of n |N(0, sigma)| samples in units of sigma -- the value the worst of n cosine errors exceeds about 63 % of
    the time (NOT its mean, which is 0.58 / a_n higher, and not a bound): 16 384 -> 4.03, 262 144 -> 4.63, 6.4e6 -> 5.26, 2.6e7 -> 5.51.
    Measured max / rms sits 1-5 % below it at every size (bench.py configs.c4).  A tolerance has to be held against a QUANTILE of the maximum:
    ``max_sigmas_quantile``."""
    return max_sigmas_gumbel(n)[1]


def max_sigmas_quantile(n: float, q: float = CONFIDENCE) -> float:
    """z with P(max_i |x_i| <= z sigma) = q over n independent N(0, sigma) samples, exactly: erfc(z / sqrt 2) = 1 - q^(1/n).  q = 0.99:
    2.64e7 -> 6.26 (the location of the maximum is 5.51), 262 144 -> 5.50, 16 384 -> 4.99; the extreme-value asymptote b_n - ln(-ln q) / a_n is
    0.02-0.05 higher.  Cosine errors of one tile against different prompts are positively correlated (one error vector, projected on the
    prompts), which only lowers the maximum: treating them as independent is the conservative side."""
    n = max(float(n), 1.0)
    q = min(max(q, 1e-300), 1.0 - 1e-15)
    p = -math.expm1(math.log(q) / n)          # 1 - q^(1/n), accurate for tiny values
    lo, hi = 0.0, 40.0
    for _ in range(200):                      # erfc is monotone: bisection to the last bit
        mid = 0.5 * (lo + hi)
        if math.erfc(mid / math.sqrt(2.0)) > p:
            lo = mid
        else:
            hi = mid
    return 0.5 * (lo + hi)


def exceedance_probability(rms: float, n: float, tolerance: float = TOLERANCE) -> float:
    """P(at least one of n independent N(0, rms) cosine errors exceeds ``tolerance``) = 1 - (1 - erfc(tolerance / (rms sqrt 2)))^n."""
    if not rms > 0.0:
        return 0.0
    tail = math.erfc(tolerance / (rms * math.sqrt(2.0)))
    if tail >= 1.0:
        return 1.0
    return float(-math.expm1(max(float(n), 1.0) * math.log1p(-tail)))


def mixture_exceedance(sigmas: Sequence[float], n: float, x: float) -> float:
    """P(at least one of n independent zero-mean Gaussian errors exceeds x in magnitude) when the errors' standard deviations are spread like
    ``sigmas`` (each value standing for n / len(sigmas) samples): 1 - prod_i (1 - erfc(x / (sigma_i sqrt 2)))^(n / len).  Tiles are not equally hard
    (a tile with more texture carries a larger error vector): the worst of a population is set by its hardest tiles, not by the rms over all."""
    sig = [s for s in sigmas if s > 0.0]
    if not sig:
        return 0.0
    w = max(float(n), 1.0) / len(sigmas)
    acc = 0.0
    for s in sig:
        tail = math.erfc(x / (s * math.sqrt(2.0)))
        if tail >= 1.0:
            return 1.0
        acc += w * math.log1p(-tail)
    return float(-math.expm1(acc))


def mixture_max_quantile(sigmas: Sequence[float], n: float, q: float = CONFIDENCE) -> float:
    """x with P(max of the n errors <= x) = q under the mixture of ``mixture_exceedance`` (bisection); for equal sigmas this is
    sigma x max_sigmas_quantile(n, q)."""
    top = max(sigmas) if len(sigmas) else 0.0
    if not top > 0.0:
        return 0.0
    lo, hi = 0.0, 40.0 * top
    for _ in range(100):
        mid = 0.5 * (lo + hi)
        if mixture_exceedance(sigmas, n, mid) > 1.0 - q:
            lo = mid
        else:
            hi = mid
    return 0.5 * (lo + hi)


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
        self._plan: Optional[Plan] = None        # per-block plan set through set_plan (re-applied whenever the handle is re-created)
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
        # plan a handle starts with (block 0 treated in full, the CLS rows' MLP redone as split products everywhere else), which no measurement on THESE weights backs.
        self.auto_calibrate = os.environ.get("KEEP_CALIBRATE", "1") != "0"
        # "measured": one split-product encode of a 64-tile probe per block and half ranks the knobs for THESE weights, then a greedy plan is verified
        # (about 2 s at load for ViT-L); "ladder": the prefix family COMP_LADDER only (under 1 s, up to 12 % slower plans -- profiles/r05_precision_budget.md)
        self.calibration_budget = os.environ.get("KEEP_CALIBRATION_BUDGET", "measured")
        self.calibration: Optional[dict] = None
        # what calibrate() / calibrate_bias() probe when the caller passes no tiles: "mixture" = N(0,1) pixels AND the structured tile families of
        # keep_amd.synth.calibration_probe, the plan being held to the WORST family; "gaussian" = N(0,1) pixels only (round 5's probe: it does not
        # hold the tolerance on structured tiles -- profiles/r06_offdist_parity_before_repair_gaussian_probe.json)
        self.calibration_probe = os.environ.get("KEEP_CALIBRATION_PROBE", "mixture")
        # load_state_dict on a GPU also runs calibrate_bias(): the mean-input compensation of the weight-rounding error that the plain fp16 launches use
        # (keep_calibrate_bias; KEEP_BIAS_CORRECTION=0 / bias_correction=False: the checkpoint's own biases everywhere)
        # Round 6: OFF by default.  The correction is exact for the mean input row of the tiles it was calibrated on and only for it: calibrated on N(0,1)
        # tiles it adds error on glass background, calibrated on the mixture probe it adds error on N(0,1) tiles, and the plan calibrate() then needs
        # is dearer with it than without (profiles/r06_offdist_parity_*.json).  A caller who knows their tiles opts in: calibrate_bias(tiles=own) + calibrate(tiles=own).
        self.bias_correction = os.environ.get("KEEP_BIAS_CORRECTION", "0") != "0"
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
        self._apply_options()

    def _apply_options(self):
        """Push the stored options, then the per-block plan (the comp_* shorthands rewrite the plan, so it goes last)."""
        lib, h = _lib.load(), self._handle
        for k, v in self._options.items():
            _lib.check(h, lib.keep_set_option(h, k.encode(), float(v)), k)
        if self._plan is not None:
            for i, (a, m) in enumerate(self._plan):
                _lib.check(h, lib.keep_set_block_precision(h, i, int(a), int(m)), "set_block_precision")

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
        self._label_margin_unit = None
        self._weights_epoch = getattr(self, "_weights_epoch", 0) + 1      # invalidates per-model prompt caches (keep_amd.wsi)
        self.calibration = None
        if self.auto_calibrate and lib.keep_vit_depth(h) > 0:
            if self.bias_correction and self._options["precision"] != _lib.PREC_STRICT:
                self.calibrate_bias()
            if self._options["precision"] == _lib.PREC_COMP:
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
            self._apply_options()
        return self

    _PLAN_SHORTHANDS = ("comp_full_blocks", "comp_mlp_blocks", "comp_qkv", "comp_qkv_from")

    def set_option(self, name: str, value: float):
        self._options[name] = value
        if name == "label_margin":
            self._label_margin_unit = None        # an explicit threshold replaces the calibrated, bank-scaled one
        if name in self._PLAN_SHORTHANDS:
            self._plan = None                 # the engine rewrites the whole per-block plan from the shorthands
        if self._handle.value:
            _lib.check(self._handle, _lib.load().keep_set_option(self._handle, name.encode(), float(value)), name)
        return self

    def set_plan(self, plan: Sequence[Tuple[int, int]]):
        """The per-block plan of the 'comp' mode: ``plan[i] = (attention-side mode, MLP mode)`` of ViT block i (``_lib.ATTN_*`` / ``_lib.MLP_*``,
        = KEEP_ATTN_* / KEEP_MLP_* of include/keep_hip.h).  Blocks beyond ``len(plan)`` run plain fp16 passes."""
        plan = [(int(a), int(m)) for a, m in plan]
        if any(not (0 <= a <= 5 and 0 <= m <= 4) for a, m in plan) or len(plan) > 64:
            raise ValueError("a plan holds at most 64 (attn_mode 0..5, mlp_mode 0..4) pairs")
        pre = plan_prefix(plan)
        for k in self._PLAN_SHORTHANDS:
            self._options.pop(k, None)
        if pre is not None:                   # a prefix plan is stored as its shorthand (and reads back through get_option); the handle's qkv
            # variants of the shorthand are reset first, so that what the engine rebuilds from the prefix is exactly `plan`
            self._options["comp_qkv"], self._options["comp_qkv_from"] = 0, 1 << 20
            self._options["comp_full_blocks"], self._options["comp_mlp_blocks"] = pre
            self._plan = None
        else:
            self._options["comp_full_blocks"] = self._options["comp_mlp_blocks"] = 0
            self._plan = plan + [(0, 0)] * (64 - len(plan))
        if self._handle.value:
            self._apply_options()
        return self

    def get_plan(self) -> Plan:
        """What the engine will run per ViT block in the 'comp' mode (read back from the handle)."""
        self._ready_device()
        lib, h = _lib.load(), self._handle
        depth = int(lib.keep_vit_depth(h)) or self.config.vision.depth
        out, a, m = [], C.c_int(0), C.c_int(0)
        for i in range(depth):
            _lib.check(h, lib.keep_get_block_precision(h, i, C.byref(a), C.byref(m)), "get_block_precision")
            out.append((a.value, m.value))
        return out

    @torch.no_grad()
    def calibrate_bias(self, tiles: Optional[torch.Tensor] = None, n_tiles: int = 64, seed: int = 20250936, probe: Optional[str] = None) -> "KEEPModel":
        """Mean-input compensation of the weight-rounding error (``keep_calibrate_bias``): the engine encodes ``tiles`` in split products, averages the
        input rows of every GEMM of the image tower and folds ``W_lo @ mean_input`` into the bias its plain fp16 launches use -- the row-independent
        part of the term a single fp16 pass drops, at no cost per call.  The correction is exact for the MEAN input row of ``tiles`` and only for it:
        pass tiles of the caller's own distribution for a closer mean.  Default (``probe="mixture"``): ``n_tiles`` tiles of
        ``keep_amd.synth.calibration_probe`` (N(0,1) pixels and the structured families in equal parts; round 6 -- calibrated on N(0,1) tiles alone
        (``probe="gaussian"``, what round 5 did) the compensation ADDS error on near-constant background tiles,
        profiles/r06_offdist_parity_before_repair_gaussian_probe.json).  ``calibrate()`` verifies whatever this leaves, family by family.  ``tiles`` of
        zero length forgets the calibration."""
        self._ready()
        dev = self._device
        if tiles is None:
            probe = probe or self.calibration_probe
            if probe == "gaussian":
                g = torch.Generator(device=dev).manual_seed(seed)
                tiles = torch.randn(n_tiles, 3, 224, 224, device=dev, generator=g, dtype=torch.float32).to(torch.bfloat16)
            elif probe == "mixture":
                from .synth import PROBE_FAMILIES, calibration_probe
                tiles = calibration_probe(max(n_tiles // len(PROBE_FAMILIES), 1), dev, seed=seed)[0]
            else:
                raise ValueError("probe must be 'mixture' or 'gaussian'")
        x = tiles.to(dev)
        if x.dtype not in _PIX and x.dtype != torch.uint8:
            x = x.to(torch.float32)
        x = x.contiguous()
        code = _lib.PIX_U8_HWC if x.dtype == torch.uint8 else _PIX[x.dtype]
        _lib.check(self._handle, _lib.load().keep_calibrate_bias(self._handle, _ptr(x) if x.shape[0] else C.c_void_p(0), code, x.shape[0], _stream(dev)), "calibrate_bias")
        self._weights_epoch = getattr(self, "_weights_epoch", 0)      # (text features do not depend on it: the prompt caches stay valid)
        return self

    @torch.no_grad()
    def calibrate(self, n_tiles: int = 1024, population: Optional[float] = None, tiles: Optional[torch.Tensor] = None,
                  text_features: Optional[torch.Tensor] = None, seed: int = 20250929, tolerance: float = TOLERANCE,
                  confidence: float = CONFIDENCE, budget: Optional[str] = None, knobs: Optional[Mapping] = None,
                  label_population: Optional[float] = None, groups: Optional[torch.Tensor] = None, group_names: Optional[Sequence[str]] = None,
                  probe: Optional[str] = None) -> Optional[dict]:
        """Pick the 'comp' plan for THESE weights and for the population the model will be used on.

        A probe batch is encoded once with split products (the engine's fp32-class arithmetic, ~5e-7 from the fp32 reference) and then with candidate
        plans from the cheapest up.  Kept: the first plan whose cosine errors over probe tiles x prompts predict, with probability ``confidence``
        (0.99), a worst error <= ``tolerance`` (1e-4) over ``population`` cosines (tiles x distinct prompts the caller will compare; default
        ``CALIBRATION_POPULATION`` = a 100 000-tile slide x the 264 distinct prompts of the RCC bank, BASELINE config 4):

            max over tile groups g of  mixture_max_quantile({sigma_t : t in g}, population, confidence) <= tolerance,
            sigma_t = (isotropic error of probe tile t) x anisotropy x margin_g x tail_g

        i.e. the ``confidence`` quantile of the largest of ``population`` Gaussian errors whose standard deviations are spread like the group's
        per-tile errors (tiles are not equally hard; for equal sigmas this is rms x max_sigmas_quantile(population, confidence)).

        The probe: ``tiles`` (with ``groups``, an int64 group index per tile, when they come from several distributions) or, by default
        (``probe="mixture"``), ``n_tiles`` tiles of ``keep_amd.synth.calibration_probe`` -- N(0,1) pixels (BASELINE config 2) and the structured
        families (stain fields, glass background, half / half) in equal parts, one group each.  The plan is held to the WORST group: a slide can be
        all tissue or mostly glass, and rounding errors behave differently there -- on spatially correlated tiles most of the error vector is COMMON
        to the tiles of a family (it does not average out across rows in attention, and the knobs that fix N(0,1) tiles leave it alone; round 6,
        profiles/r06_offdist_parity_*.json).  ``probe="gaussian"`` = round 5's probe.

        ``rms_g`` = the group's ISOTROPIC rms |feature error| / sqrt(D) -- what the rms would be against prompts in random directions -- times one
        anisotropy factor for the plan: the median over the groups of (rms against the probe's prompt bank / isotropic rms), at least 1 (error
        vectors are not isotropic; but the bank rms of a group of near-identical tiles is one random draw, so no single group's ratio is trusted).
        ``margin_g`` >= PROBE_RMS_MARGIN = two standard errors of the group's mean squared error, estimated from its tile-to-tile spread (one tile's
        cosines share ONE error vector).  ``tail_g`` >= 1 only when the group's own maximum is larger than its per-tile mixture allows a sample of the
        probe's size at 99 %.  ``model.calibration`` reports the prediction, the per-group figures and the ``exceedance_probability`` of the chosen plan.
        Candidates: ``budget="ladder"`` walks ``COMP_LADDER`` (prefix plans); ``budget="measured"`` first measures, on these weights and per group,
        the variance share of every block's attention side and MLP (one split-product encode per block and half with that one site downgraded),
        then builds the plan greedily -- each step the upgrade that lowers the worst group's predicted variance most per millisecond
        (``KNOB_COST_MS``) -- and verifies it the same way.  The prompts are ``text_features`` ([P,768] unit rows -- pass the caller's own bank to
        calibrate against it; default: 64 seeded prompts through the loaded text tower, or 64 seeded random unit vectors for an image-only engine,
        a harsher yardstick).  If nothing qualifies the engine switches to 'strict'.  ``label_margin`` (keep_classify's second-look threshold) is set
        from the measured rms.  Non-finite probe features (an activation beyond the fp16 range) raise FloatingPointError.  ``strict_blocks`` is
        kept; on an exception the previous setting is restored.  Returns and stores ``self.calibration``."""
        lib, h = _lib.load(), self._handle
        if not self._loaded and self._host_sd is not None:
            self.to("cuda")
            if self.calibration is not None:
                return self.calibration
        if not h.value or lib.keep_vit_depth(h) == 0 or self._options["precision"] != _lib.PREC_COMP:
            return None
        budget = budget or self.calibration_budget
        if budget not in ("ladder", "measured"):
            raise ValueError("budget must be 'ladder' or 'measured'")
        dev, depth = self._device, int(lib.keep_vit_depth(h))
        population = float(CALIBRATION_POPULATION if population is None else population)
        probe_name = "caller's tiles"
        if tiles is None:
            probe_name = probe or self.calibration_probe
            if probe_name == "gaussian":
                g = torch.Generator(device=dev).manual_seed(seed)
                tiles = torch.randn(n_tiles, 3, 224, 224, device=dev, generator=g, dtype=torch.float32).to(torch.bfloat16)
                groups, group_names = None, ("gaussian",)
            elif probe_name == "mixture":
                from .synth import PROBE_FAMILIES, calibration_probe
                tiles, groups, group_names = calibration_probe(max(n_tiles // len(PROBE_FAMILIES), 1), dev, seed=seed)
            else:
                raise ValueError("probe must be 'mixture' or 'gaussian'")
        tiles = tiles.to(dev)
        n_t = int(tiles.shape[0])
        if groups is None:
            groups = torch.zeros(n_t, dtype=torch.int64)
        groups = torch.as_tensor(groups, dtype=torch.int64).cpu()
        if groups.numel() != n_t or (n_t and int(groups.min()) < 0):
            raise ValueError("groups: one non-negative group index per probe tile")
        n_groups = int(groups.max()) + 1 if n_t else 1
        names = tuple(group_names) if group_names is not None else tuple(f"group{i}" for i in range(n_groups))
        if len(names) < n_groups:
            raise ValueError("group_names: one name per group index")
        members = [torch.nonzero(groups == gi).flatten().to(dev) for gi in range(n_groups)]
        members = [(names[gi], m) for gi, m in enumerate(members) if m.numel() > 0]
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
        z_pop = max_sigmas_quantile(population, confidence)
        rms_target = tolerance / (z_pop * PROBE_RMS_MARGIN)
        saved_opts, saved_plan = dict(self._options), (list(self._plan) if self._plan is not None else None)
        strict_blocks = int(saved_opts.get("strict_blocks") or 0)
        was, self.auto_calibrate = self.auto_calibrate, False
        tried, chosen, chosen_stats, shares = [], None, None, None
        done = False

        def encode_probe(x):
            return torch.cat([self.encode_image(x[i:i + 256]) for i in range(0, x.shape[0], 256)]) if x.shape[0] > 256 else self.encode_image(x)

        def probe_stats(plan: Plan):
            """Encode the probe under `plan`; per group: max, isotropic and bank rms, margin, tail -> the governing group's figures."""
            self.set_plan(plan)
            f = encode_probe(tiles)
            d2 = self.similarity(f, bank).sub_(ref).pow_(2)                    # [n, P] squared cosine errors
            e2 = (f - ref_f).pow(2).sum(dim=1).div_(f.shape[1])               # [n] isotropic squared error per tile
            rows = []
            for name, idx in members:
                dg, eg = d2[idx], e2[idx]
                nt = int(idx.numel())
                mx = float(dg.max().sqrt())
                rms_bank, ms_iso = math.sqrt(float(dg.mean())), float(eg.mean())
                rms_iso = math.sqrt(ms_iso)
                # relative standard error of the group's mean squared error from its tile-to-tile spread (one tile's cosines share ONE error vector:
                # the tile is the sample); the rule holds the plan to two of them, at least PROBE_RMS_MARGIN
                se = float(eg.std()) / math.sqrt(nt) / ms_iso if nt > 1 and ms_iso > 0 else 0.0
                margin = max(PROBE_RMS_MARGIN, math.sqrt(1.0 + 2.0 * se))
                rows.append([name, mx, rms_bank, rms_iso, margin, 1.0, nt])
            # error vectors are not isotropic: against a bank of real prompts the rms can sit 5-10 % above (or below) the isotropic figure.  ONE
            # factor for the plan, the MEDIAN over the groups of bank rms / isotropic rms, at least 1: a group of near-identical tiles (glass) has
            # near-identical error vectors, and against a bank of similar prompts its bank rms is a single random draw, not an rms
            ratios = sorted(r[2] / r[3] for r in rows if r[3] > 0)
            aniso = max(1.0, 0.5 * (ratios[(len(ratios) - 1) // 2] + ratios[len(ratios) // 2])) if ratios else 1.0
            per_group, worst = {}, None
            for (name, mx, rms_bank, rms_iso, margin, tail, nt), (_, idx) in zip(rows, members):
                rms = rms_iso * aniso
                # heavier tails than the model assumes show as a probe maximum above what the SAME per-tile mixture allows a sample of the probe's
                # size at 99 %: then every sigma is stretched by the ratio (measured against the isotropic per-tile figures, not against the
                # group's bank rms -- for a group of near-identical tiles that is a single random draw)
                allowed = mixture_max_quantile((e2[idx].double().sqrt() * aniso).tolist(), nt * bank.shape[0], 0.99)
                tail = max(1.0, mx / allowed) if allowed > 0 else 1.0
                # the group's tiles are not equally hard: the population maximum is predicted from the per-tile spread (mixture_max_quantile),
                # every tile's isotropic error standing for population / nt cosines; `eff` = the homoscedastic rms that predicts the same maximum
                sig = (e2[idx].double().sqrt() * (aniso * margin * tail)).tolist()
                pred_g = mixture_max_quantile(sig, population, confidence)
                eff = pred_g / z_pop
                per_group[name] = {"max_abs_dcos": float(f"{mx:.3e}"), "rms_dcos_vs_bank": float(f"{rms_bank:.3e}"), "rms_dcos_isotropic": float(f"{rms_iso:.3e}"),
                                   "margin": round(margin, 4), "tail_factor": round(tail, 3), "effective_rms": float(f"{eff:.3e}"),
                                   "hardest_tile_over_rms": round(max(sig) / (rms * margin * tail), 3) if rms > 0 else 0.0, "tiles": nt,
                                   "exceedance_probability": float(f"{mixture_exceedance(sig, population, tolerance):.3e}")}
                if worst is None or eff > worst[1]:
                    worst = (name, eff, mx, rms, margin, tail, rms_bank, rms_iso, sig)
            per_group["anisotropy_factor"] = round(aniso, 4)
            err_all = max(r[1] for r in rows)
            return err_all, worst, per_group

        def consider(plan: Plan) -> bool:
            nonlocal chosen, chosen_stats
            err, worst, per_group = probe_stats(plan)
            name, eff, _, rms, margin, tail, rms_bank, rms_iso, sig = worst
            pred = eff * z_pop
            pre = plan_prefix(plan)
            tried.append({"comp_full_blocks": pre[0] if pre else None, "comp_mlp_blocks": pre[1] if pre else None, "plan": plan_string(plan),
                          "max_abs_dcos": float(f"{err:.3e}"), "rms_dcos": float(f"{rms:.3e}"), "rms_dcos_vs_bank": float(f"{rms_bank:.3e}"),
                          "rms_dcos_isotropic": float(f"{rms_iso:.3e}"), "governing_group": name, "predicted_max_abs_dcos": float(f"{pred:.3e}"),
                          "isotropic_rms_by_group": {k: v["rms_dcos_isotropic"] for k, v in per_group.items() if isinstance(v, dict)},
                          "anisotropy_factor": per_group["anisotropy_factor"]})
            if pred <= tolerance:                 # (NaN compares False: falls through to the next candidate)
                chosen, chosen_stats = plan, (err, rms, pred, tail, margin, name, per_group, sig)
                return True
            return False

        try:
            self.set_precision("strict", strict_blocks)
            ref_f = encode_probe(tiles)
            ref = self.similarity(ref_f, bank)                              # cosines on the engine's exact-fp32 similarity kernel
            self.set_precision("comp", strict_blocks)
            if not bool(torch.isfinite(ref).all()):
                self._raise_flags(2)
            if budget == "ladder":
                for full, mlp in dict.fromkeys((min(a, depth), min(b, depth)) for a, b in COMP_LADDER):
                    if consider(prefix_plan(depth, full, mlp)):
                        break
            else:
                # the shares only rank the knobs: a sixteenth of the probe (at least 16 tiles of every group) is enough
                sel = torch.cat([idx[:max(16, int(idx.numel()) // 16)] for _, idx in members])
                sh_members = []
                at = 0
                for name, idx in members:
                    k = min(max(16, int(idx.numel()) // 16), int(idx.numel()))
                    sh_members.append((name, torch.arange(at, at + k, device=dev)))
                    at += k
                shares = self._measure_shares(tiles[sel], ref_f[sel], depth, sh_members)
                walk = self._greedy_walk(shares, depth, knobs or DEFAULT_KNOBS)
                # the sum of measured shares over-predicts the rms of a plan by 3-11 % (variances of neighbouring sites do not quite add): start the
                # verification a little before the predicted crossing and walk up one knob at a time until a plan verifies
                start = next((i for i, (_, v) in enumerate(walk) if v <= (1.06 * rms_target) ** 2), len(walk) - 1)
                # verified from there upwards, ONE knob at a time, until a plan passes.  (Not by bisection: the prediction is not monotone along the walk --
                # the anisotropy factor moves between 1.0 and 2.0 once the isotropic error is small -- and a bisection that lands in that region keeps a
                # plan several times dearer than the first one that verifies: round 6 saw 47.8 instead of 35.4 ms per step.)
                for plan, _ in walk[start:]:
                    if consider(plan):
                        break
            if chosen is None:
                self.set_precision("strict", strict_blocks)
            self.check_errors(wait=True)
            done = True
        finally:
            self.auto_calibrate = was
            if not done:                 # an exception mid-way: put back exactly what the caller had
                self._options, self._plan = saved_opts, saved_plan
                if self._handle.value:
                    try:
                        self._apply_options()
                    except Exception:
                        pass
        pre = plan_prefix(chosen) if chosen else None
        self.calibration = {"precision": "comp" if chosen else "strict", "comp_full_blocks": pre[0] if pre else None,
                            "comp_mlp_blocks": pre[1] if pre else None, "plan": plan_string(chosen) if chosen else None, "budget": budget,
                            "population": population, "confidence": confidence, "max_sigmas_quantile": round(z_pop, 3),
                            "expected_max_sigmas": round(expected_max_sigmas(population), 3),
                            "target_rms_dcos": float(f"{rms_target:.3e}"), "strict_blocks": strict_blocks,
                            "probe": f"{tiles.shape[0]} tiles x {bank.shape[0]} prompts vs the split-product arithmetic", "probe_distribution": probe_name,
                            "probe_groups": [name for name, _ in members], "tried": tried,
                            "bias_correction": bool(lib.keep_get_option(h, b"bias_ready") > 0 and lib.keep_get_option(h, b"bias_correction") > 0)}
        if chosen:
            err, rms, pred, tail, margin_g, gov, per_group, sig = chosen_stats
            # keep_classify looks a second time at tiles whose top-2 cosine margin could hide a flipped label.  A margin is the difference of two
            # cosines of ONE tile, i.e. its error vector projected on t1 - t2: standard deviation |t1 - t2| x rms, and there is ONE margin per tile:
            # the quantile is taken over the tiles of the population (`label_population`; default: the 100 000 tiles of the default population, the
            # whole population for a caller's own), at the same confidence.  |t1 - t2| <= 2 in general; the engine's default is sqrt 2 (prompts
            # with a non-negative cosine -- every pair of the banks this repository has seen); ``classify`` scales it by the caller's own bank.
            n_margins = float(label_population) if label_population else (100_000.0 if population == float(CALIBRATION_POPULATION) else population)
            unit = mixture_max_quantile(sig, n_margins, confidence)
            margin = math.sqrt(2.0) * unit
            self.set_option("label_margin", margin)
            self._label_margin_unit = unit
            self.calibration.update({"probe_max_abs_dcos": float(f"{err:.3e}"), "probe_rms_dcos": float(f"{rms:.3e}"), "tail_factor": round(tail, 3),
                                     "rms_margin": round(margin_g, 4), "governing_group": gov, "per_group": per_group,
                                     "predicted_max_abs_dcos": float(f"{pred:.3e}"),
                                     "exceedance_probability": float(f"{mixture_exceedance(sig, population, tolerance):.3e}"),
                                     "label_margin": float(f"{margin:.3e}"), "label_margin_per_unit_prompt_distance": float(f"{unit:.3e}"),
                                     "label_population": n_margins})
        if shares is not None:
            self.calibration["variance_shares"] = shares
        return self.calibration

    def _measure_shares(self, tiles, ref_f, depth: int, members=None) -> dict:
        """Cosine-error variance each block's attention side / MLP contributes when it alone runs single fp16 passes and everything else split
        products (so the figure is that site's own rounding error, not the re-drawn rounding of everything downstream), plus what is left of a
        site's share under the cheaper treatments -- per tile group (``members``: [(name, indices into ``tiles``)]).  Variances are the isotropic
        ones -- |feature error|^2 / D, the mean squared cosine error over random prompt directions -- which ranks the knobs independently of any
        prompt bank.  Top-level ``attn`` / ``mlp`` / ``floor`` / ``residual_*`` hold the worst group's value per entry; ``by_group`` everything."""
        split = [(_lib.ATTN_SPLIT, _lib.MLP_SPLIT)] * depth
        if members is None:
            members = [("all", torch.arange(tiles.shape[0], device=tiles.device))]
        G = len(members)

        def iso_var():
            f = self.encode_image(tiles) if tiles.shape[0] <= 256 else torch.cat([self.encode_image(tiles[i:i + 256]) for i in range(0, tiles.shape[0], 256)])
            e2 = (f - ref_f).pow(2).sum(dim=1).div_(f.shape[1])
            return [float(e2[idx].mean()) for _, idx in members]

        def var_of(i, mode):
            p = list(split)
            p[i] = mode
            self.set_plan(p)
            return iso_var()

        def minus_floor(v):
            return [max(v[g] - floor[g], 0.0) for g in range(G)]

        def ratio(v, base):
            return [min(v[g] / base[g], 1.0) if base[g] > 0 else 0.0 for g in range(G)]

        self.set_plan(split)
        floor = iso_var()
        attn = [minus_floor(var_of(i, (_lib.ATTN_PLAIN, _lib.MLP_SPLIT))) for i in range(depth)]          # [depth][G]
        mlp = [minus_floor(var_of(i, (_lib.ATTN_SPLIT, _lib.MLP_PLAIN))) for i in range(depth)]
        res_m = {k: [v] * G for k, v in MLP_RESIDUAL.items()}
        res_a = {k: [v] * G for k, v in ATTN_RESIDUAL.items()}
        # what the CLS-rows-only treatment leaves differs from block to block (most in the first, where every row's error is amplified by all the
        # attention layers that follow) and from group to group (on correlated tiles the other rows carry the SAME error): measured for every block
        cls_left = [ratio(minus_floor(var_of(i, (_lib.ATTN_SPLIT, _lib.MLP_CLS))), mlp[i]) for i in range(depth)]
        for mode in (_lib.MLP_COMP, _lib.MLP_COMP_W):
            fr = [ratio(minus_floor(var_of(i, (_lib.ATTN_SPLIT, mode))), mlp[i]) for i in sorted({0, depth // 2})]
            res_m[mode] = [sum(f[g] for f in fr) / len(fr) for g in range(G)]
        for mode in (_lib.ATTN_SPLIT_COMPQKV, _lib.ATTN_COMPQKV, _lib.ATTN_COMPQKV_PROJ_CLS):
            fr = [ratio(minus_floor(var_of(i, (mode, _lib.MLP_SPLIT))), attn[i]) for i in sorted({0, min(1, depth - 1), depth // 2})]
            res_a[mode] = [max(f[g] for f in fr) for g in range(G)]
        # ... and so does what the CLS-row proj leaves of an attention side (nearly all of it in block 0 of N(0,1) tiles, under half on correlated tiles)
        pcls_left = [ratio(minus_floor(var_of(i, (_lib.ATTN_PROJ_CLS, _lib.MLP_SPLIT))), attn[i]) for i in range(depth)]
        r3 = lambda v: float(f"{v:.3e}")
        r4 = lambda v: float(f"{v:.4f}")
        by_group = {}
        for g, (name, idx) in enumerate(members):
            rm = {int(k): r4(v[g]) for k, v in res_m.items()}
            rm[int(_lib.MLP_CLS)] = [r4(cls_left[i][g]) for i in range(depth)]
            ra = {int(k): r4(v[g]) for k, v in res_a.items()}
            ra[int(_lib.ATTN_PROJ_CLS)] = [r4(pcls_left[i][g]) for i in range(depth)]
            by_group[name] = {"attn": [r3(attn[i][g]) for i in range(depth)], "mlp": [r3(mlp[i][g]) for i in range(depth)], "floor": r3(floor[g]),
                              "residual_mlp": rm, "residual_attn": ra, "probe_tiles": int(idx.numel())}
        worst_m = {int(k): r4(max(v)) for k, v in res_m.items()}
        worst_m[int(_lib.MLP_CLS)] = [r4(max(cls_left[i])) for i in range(depth)]
        worst_a = {int(k): r4(max(v)) for k, v in res_a.items()}
        worst_a[int(_lib.ATTN_PROJ_CLS)] = [r4(max(pcls_left[i])) for i in range(depth)]
        return {"attn": [r3(max(attn[i])) for i in range(depth)], "mlp": [r3(max(mlp[i])) for i in range(depth)], "floor": r3(max(floor)),
                "residual_mlp": worst_m, "residual_attn": worst_a, "probe_tiles": int(tiles.shape[0]),
                "groups": [name for name, _ in members], "by_group": by_group}

    @staticmethod
    def _greedy_walk(shares: dict, depth: int, knobs: Mapping) -> List[Tuple[Plan, float]]:
        """[(plan, predicted cosine-error variance of the WORST tile group)] from the all-plain plan upwards: every step takes the ONE upgrade (a
        block's attention side or MLP to one of the allowed treatments) that lowers the worst group's predicted variance most per millisecond
        (KNOB_COST_MS); when no upgrade moves the worst group any more, the one with the largest summed reduction.  The last entry has every site
        at its best allowed treatment.  ``shares`` without ``by_group`` (one group) are read from the top level."""
        groups = list(shares["by_group"].values()) if shares.get("by_group") else [shares]
        G = len(groups)
        res_a = [{int(k): v for k, v in g["residual_attn"].items()} for g in groups]
        res_m = [{int(k): v for k, v in g["residual_mlp"].items()} for g in groups]
        left_m = lambda g, i, mode: (res_m[g][mode][i] if isinstance(res_m[g][mode], (list, tuple)) else res_m[g][mode])
        left_a = lambda g, i, mode: (res_a[g][mode][i] if isinstance(res_a[g][mode], (list, tuple)) else res_a[g][mode])
        cost_a = {_lib.ATTN_PLAIN: 0.0, _lib.ATTN_PROJ_CLS: KNOB_COST_MS["attn_proj_cls"], _lib.ATTN_COMPQKV: KNOB_COST_MS["attn_compqkv"],
                  _lib.ATTN_COMPQKV_PROJ_CLS: KNOB_COST_MS["attn_compqkv_proj_cls"],
                  _lib.ATTN_SPLIT_COMPQKV: KNOB_COST_MS["attn_split_compqkv"], _lib.ATTN_SPLIT: KNOB_COST_MS["attn_split"]}
        cost_m = {_lib.MLP_PLAIN: 0.0, _lib.MLP_CLS: KNOB_COST_MS["mlp_cls"], _lib.MLP_COMP_W: KNOB_COST_MS["mlp_comp_w"], _lib.MLP_COMP: KNOB_COST_MS["mlp_comp"],
                  _lib.MLP_SPLIT: 3.0 * KNOB_COST_MS["mlp_comp"]}
        am, mm = [_lib.ATTN_PLAIN] * depth, [_lib.MLP_PLAIN] * depth

        def predicted():
            return [groups[g]["floor"] + sum(groups[g]["attn"][i] * left_a(g, i, am[i]) + groups[g]["mlp"][i] * left_m(g, i, mm[i]) for i in range(depth))
                    for g in range(G)]

        cur = predicted()
        walk = [(list(zip(am, mm)), max(cur))]
        while True:
            best, gain, best_sum, gain_sum = None, 0.0, None, 0.0
            top = max(cur)
            for i in range(depth):
                cands = [(am, a, cost_a[a] - cost_a[am[i]], [groups[g]["attn"][i] * (left_a(g, i, am[i]) - left_a(g, i, a)) for g in range(G)]) for a in knobs.get("attn", ())
                         if a in res_a[0]]
                cands += [(mm, m, cost_m[m] - cost_m[mm[i]], [groups[g]["mlp"][i] * (left_m(g, i, mm[i]) - left_m(g, i, m)) for g in range(G)]) for m in knobs.get("mlp", ())]
                for target, mode, dc, dv in cands:
                    if dc <= 0:
                        continue
                    drop = top - max(cur[g] - dv[g] for g in range(G))
                    if drop > 0 and drop / dc > gain:
                        best, gain = (target, i, mode), drop / dc
                    if sum(dv) > 0 and sum(dv) / dc > gain_sum:
                        best_sum, gain_sum = (target, i, mode), sum(dv) / dc
            best = best or best_sum
            if best is None:
                break
            best[0][best[1]] = best[2]
            cur = predicted()
            walk.append((list(zip(am, mm)), max(cur)))
        return walk

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
        tiles whose two best prompts are closer than ``margin`` in cosine are encoded a second time with split products (``strict``) and
        take their row from that.  Default margin: what ``calibrate()`` measured -- the worst error one tile's cosine DIFFERENCE can carry at the
        calibration's confidence = (per-unit figure ``calibration["label_margin_per_unit_prompt_distance"]``) x the largest distance |t_i - t_j|
        between two prompts of ``text_features`` (at most 2; sqrt 2 for banks without negative cosines, which is what the engine's own
        ``label_margin`` option assumes for C-ABI callers); an uncalibrated handle starts at 2.5e-4 = twice the 1e-4 tolerance + 25 %.
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
        if margin is None and getattr(self, "_label_margin_unit", None):
            margin = self._label_margin_unit * self._bank_diameter(txt)
        self.last_margin = float(margin) if margin is not None else (self.get_option("label_margin") if self._handle.value else None)
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

    def _bank_diameter(self, txt: torch.Tensor) -> float:
        """max |t_i - t_j| over the rows of a prompt bank (unit rows: sqrt(2 - 2 min cos)), on the engine's own similarity kernel; cached for the
        last bank.  Banks too large for a P x P matrix (or a single prompt) take the bound 2."""
        P = int(txt.shape[0])
        if P < 2 or P > 8192:
            return 2.0
        key = (txt.data_ptr(), P, txt._version)
        if getattr(self, "_bank_diam_key", None) != key:
            cos_min = float(self.similarity(txt, txt).min())
            self._bank_diam_key, self._bank_diam = key, math.sqrt(max(2.0 - 2.0 * cos_min, 0.0))
        return max(self._bank_diam, 1e-3)

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

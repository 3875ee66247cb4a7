"""The L4 call surface is the reference's: same function names, positional arguments and defaults as
WSI_evaluation/{utils,subtyping_utils,detection_utils,segment_utils}.py (plus keyword-only-in-practice extras that default to
None).  The expected signatures were read off the reference (file:line in the table); when /root/reference is present (build
container) they are additionally compared with the live source."""
import inspect
import os
import sys
import types

import pytest

from keep_amd.wsi_evaluation import detection_utils, segment_utils, subtyping_utils, utils

# module, function, reference line, positional parameters with defaults as the reference declares them
EXPECTED = [
    (utils, "zero_shot_classifier", "utils.py:64", [("KEEP_model",), ("classnames",), ("templates",), ("device",)]),
    (utils, "get_zeroshot_classifier", "utils.py:86", [("model",), ("label_map",), ("prompts",), ("device",), ("add_normal", False)]),
    (utils, "rank_cls_score", "utils.py:107", [("logits",)]),
    (utils, "zero_shot_prompt_select", "utils.py:119", [("classifiers",), ("tile_features",), ("topn",), ("device",)]),
    (utils, "cood2str", "utils.py:148", [("cood",)]),
    (utils, "accuracy", "utils.py:153", [("logits",), ("target",), ("topk", (1,))]),
    (subtyping_utils, "run", "subtyping_utils.py:13", [("classifier",), ("dataloader",), ("device",)]),
    (subtyping_utils, "refine_seg", "subtyping_utils.py:38", [("logits_slide",), ("coords_slide",), ("patch_size", 224), ("overlap", True)]),
    (subtyping_utils, "zero_shot_subtyping", "subtyping_utils.py:67", [("classifier",), ("tile_features",), ("tile_coords",), ("patch_size", 256), ("overlap", True)]),
    (detection_utils, "run", "detection_utils.py:13", [("classifier",), ("dataloader",), ("device",)]),
    (detection_utils, "refine_seg", "detection_utils.py:39", [("logits_slide",), ("coords_slide",), ("patch_size", 224), ("threshold", 0.5), ("overlap", True)]),
    (detection_utils, "zero_shot_detection", "detection_utils.py:88", [("classifier",), ("tile_features",), ("tile_coords",), ("patch_size", 256), ("overlap", False)]),
    (segment_utils, "run", "segment_utils.py:17", [("classifier",), ("dataloader",), ("device",)]),
    (segment_utils, "zero_shot_segment", "segment_utils.py:44", [("classifier",), ("tile_features",), ("tile_coords",), ("mask_path",), ("patch_size", 224), ("overlap", True)]),
    (segment_utils, "refine_seg", "segment_utils.py:63", [("logits_slide",), ("coords_slide",), ("patch_size", 224), ("overlap", True)]),
]


def _leading(fn, n):
    out = []
    for p in list(inspect.signature(fn).parameters.values())[:n]:
        out.append((p.name,) if p.default is inspect.Parameter.empty else (p.name, p.default))
    return out


@pytest.mark.parametrize("mod,name,where,params", EXPECTED, ids=[f"{m.__name__.rsplit('.', 1)[-1]}.{n}" for m, n, _, _ in EXPECTED])
def test_signature_is_the_reference_one(mod, name, where, params):
    fn = getattr(mod, name)
    assert _leading(fn, len(params)) == params, where
    extra = list(inspect.signature(fn).parameters.values())[len(params):]
    assert all(p.default is None for p in extra), f"{name}: extra parameters must be optional (default None): {extra}"


@pytest.mark.skipif(not os.path.isdir("/root/reference/WSI_evaluation"), reason="reference tree only exists in the build container")
def test_expected_table_matches_the_live_reference():
    sys.modules.setdefault("h5py", types.ModuleType("h5py"))
    sys.modules.setdefault("openslide", types.ModuleType("openslide"))
    saved = {k: sys.modules.pop(k, None) for k in ("utils", "subtyping_utils", "detection_utils", "segment_utils")}
    sys.path.insert(0, "/root/reference/WSI_evaluation")
    try:
        import importlib
        ref = {n: importlib.import_module(n) for n in ("utils", "subtyping_utils", "detection_utils", "segment_utils")}
        for mod, name, where, params in EXPECTED:
            rfn = getattr(ref[mod.__name__.rsplit(".", 1)[-1]], name)
            assert _leading(rfn, len(params)) == params and len(inspect.signature(rfn).parameters) == len(params), where
    finally:
        sys.path.remove("/root/reference/WSI_evaluation")
        for k, v in saved.items():
            sys.modules.pop(k, None)
            if v is not None:
                sys.modules[k] = v

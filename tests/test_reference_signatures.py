"""The L4 call surface is the reference's: same function names, positional parameters and defaults as
WSI_evaluation/{utils,subtyping_utils,detection_utils,segment_utils}.py.  tests/golden/reference_signatures.json was produced
by `inspect.signature` on the reference's live functions (tools/make_golden.py signatures); keep_amd may only ADD optional
parameters that default to None (the engine pin `model=`, the prompt `cache=`)."""
import importlib
import inspect
import json
import os

import pytest

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_signatures.json")) as _f:
    TABLE = json.load(_f)


def _norm(v):
    return list(v) if isinstance(v, tuple) else v


@pytest.mark.parametrize("qualname", sorted(TABLE))
def test_signature_is_the_reference_one(qualname):
    mod, name = qualname.split(".")
    fn = getattr(importlib.import_module(f"keep_amd.wsi_evaluation.{mod}"), name)
    ref = TABLE[qualname]["params"]
    got = list(inspect.signature(fn).parameters.values())
    lead = [[p.name] if p.default is inspect.Parameter.empty else [p.name, _norm(p.default)] for p in got[:len(ref)]]
    assert lead == ref, TABLE[qualname]["line"]
    assert all(p.default is None for p in got[len(ref):]), f"{qualname}: extra parameters must be optional (default None)"


def test_every_reference_function_of_the_path_is_exported():
    assert len(TABLE) == 16
    for mod in ("utils", "subtyping_utils", "detection_utils", "segment_utils"):
        m = importlib.import_module(f"keep_amd.wsi_evaluation.{mod}")
        assert all(hasattr(m, q.split(".")[1]) for q in TABLE if q.startswith(mod + "."))
    from keep_amd.wsi_evaluation.utils import WSI_Classification_Dataset          # utils.py:11
    assert [p for p in inspect.signature(WSI_Classification_Dataset.__init__).parameters][1:] == \
        ["df", "data_source", "target_transform", "index_col", "target_col", "use_h5", "label_map"]

"""tools/pin_against_timm.py cannot meet timm in the build container; its plumbing (fixture regeneration from seeds, strict load of the
release's ``visual.*`` keys into the model ``timm.create_model`` returns, the 1e-6 assertion, the ``features_timm`` key the golden tests
then pick up) is exercised here against a stand-in module named ``timm`` whose ``create_model`` returns the ATen-op restatement of timm's
module tree from tools/make_golden.py.  The stand-in pins nothing: the fixture this test writes goes to a temporary directory."""
import importlib.util
import os
import shutil
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_without_timm_nothing_is_written(golden_dir, monkeypatch):
    monkeypatch.setitem(sys.modules, "timm", None)                       # `import timm` raises ImportError
    pin = _load("pin_against_timm")
    before = os.path.getmtime(os.path.join(golden_dir, "vit_d2.npz"))
    assert pin.main() == 3
    assert os.path.getmtime(os.path.join(golden_dir, "vit_d2.npz")) == before


def test_pin_plumbing_against_a_stand_in(golden_dir, tmp_path, monkeypatch):
    mg, pin = _load("make_golden"), _load("pin_against_timm")
    seen = {}

    def create_model(name, **kw):
        seen.update(kw, name=name)
        return mg._TimmViT(kw.get("depth", 24))

    fake = types.ModuleType("timm")
    fake.create_model, fake.__version__ = create_model, "stand-in"
    for vit in ("num_features",):
        setattr(mg._TimmViT, vit, 1024)
    shutil.copy(os.path.join(golden_dir, "vit_d2.npz"), tmp_path / "vit_d2.npz")
    monkeypatch.setattr(pin, "GOLD", str(tmp_path))
    d = pin.pin(fake, "vit_d2.npz")
    assert d < 1e-6
    # the constructor arguments are the reference's (quick_start/keep_inference.py:32-40), plus depth for the two-block fixture
    assert seen == dict(name="vit_large_patch16_224", pretrained=False, img_size=224, patch_size=16, init_values=1e-5, num_classes=0,
                        dynamic_img_size=True, depth=2)
    g = np.load(tmp_path / "vit_d2.npz")
    assert "features_timm" in g.files and str(g["timm_version"]) == "stand-in"
    assert np.abs(g["features_timm"] - g["features_aten_timm"]).max() < 1e-6
    assert set(np.load(os.path.join(golden_dir, "vit_d2.npz")).files) <= set(g.files)      # nothing of the fixture is lost


def test_a_disagreeing_timm_is_refused(golden_dir, tmp_path, monkeypatch):
    mg, pin = _load("make_golden"), _load("pin_against_timm")

    class Off(mg._TimmViT):
        num_features = 1024

        def forward(self, x):
            return super().forward(x) * 1.01 + 0.01                                        # not the oracle's arithmetic

    fake = types.ModuleType("timm")
    fake.create_model, fake.__version__ = (lambda name, **kw: Off(kw.get("depth", 24))), "off"
    shutil.copy(os.path.join(golden_dir, "vit_d2.npz"), tmp_path / "vit_d2.npz")
    monkeypatch.setattr(pin, "GOLD", str(tmp_path))
    with pytest.raises(SystemExit):
        pin.pin(fake, "vit_d2.npz")
    assert "features_timm" not in np.load(tmp_path / "vit_d2.npz").files

"""The C-ABI library builds, loads and exports every symbol include/keep_hip.h declares.
No compute calls: this runs in the GPU-less build container."""
import ctypes
import os
import re

import pytest

from keep_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build(verbose=False)
    return _lib.load()


def header_symbols():
    hdr = open(os.path.join(ROOT, "include", "keep_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(keep_[a-z0-9_]+)\s*\(", hdr)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = header_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in keep_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names


def test_version_and_null_handle_errors(lib):
    assert b"gfx950" in lib.keep_version()
    assert lib.keep_last_error(None) == b"null handle"
    assert lib.keep_encode_image(None, None, 0, 1, None, None) == _lib.KEEP_EINVAL
    assert lib.keep_destroy(None) == _lib.KEEP_OK


def test_code_object_is_gfx950_only():
    so = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in so
    for other in (b"gfx942", b"gfx90a", b"sm_90"):
        assert other not in so


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.KeepHipError):
        _lib.load()


def test_no_gpu_means_no_compute():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from keep_amd import KEEPModel
    from keep_amd.synth import synth_tiles
    m = KEEPModel()
    with pytest.raises(_lib.KeepHipError):
        m.encode_image(synth_tiles(1))
    with pytest.raises(_lib.KeepHipError):
        m.to("cuda")

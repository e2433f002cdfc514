"""CPU: the C-ABI library loads and exports every symbol include/faceid.h declares."""
import os
import re

from conftest import ROOT
from scrfd_arcface_facerecognition_amd import _lib


def header_symbols():
    src = open(os.path.join(ROOT, "include", "faceid.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fid_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), s
    assert sorted(_lib.SIGNATURES) == syms          # the binding covers exactly the header
    assert lib.fid_abi_version() == 1


def test_device_count_call_is_safe_without_gpu():
    assert _lib.device_count() >= 0


def test_product_does_not_import_oracle():
    """oracle/ is test infrastructure: nothing under the product package (or the drop-in shims) may use it."""
    import glob
    pat = re.compile(r"^\s*(from\s+oracle|import\s+oracle)", re.M)
    files = glob.glob(os.path.join(ROOT, "scrfd_arcface_facerecognition_amd", "**", "*.py"), recursive=True)
    files += glob.glob(os.path.join(ROOT, "models", "*.py")) + glob.glob(os.path.join(ROOT, "utils", "*.py"))
    assert len(files) > 8
    for f in files:
        assert not pat.search(open(f).read()), f


def test_library_missing_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    import pytest
    with pytest.raises(FileNotFoundError):
        _lib.load()

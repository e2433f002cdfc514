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

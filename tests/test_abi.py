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
    assert lib.fid_abi_version() == 2


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


def test_integration_md_names_resolve():
    """INTEGRATION.md is part of the boundary: every `fid_*` name in it is a declared + bound entry point (or one of
    the handle / struct types), every entry point of the header appears in it, every `module.attr` / `Class.method` of the package it
    names exists, and the calls it shows match the real signatures (it once showed a Communicator API that did
    not exist)."""
    import importlib
    import inspect
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    syms = set(header_symbols())
    types = {"fid_ctx", "fid_net", "fid_gallery", "fid_gate_config", "fid_comm"}
    named = set(re.findall(r"\bfid_[a-z0-9_]+\b", doc))
    assert named - types <= syms, sorted(named - types - syms)
    assert syms <= named, sorted(syms - named)                       # ... and the document's table covers the whole header
    assert all(n in _lib.SIGNATURES for n in named - types)
    pkg = "scrfd_arcface_facerecognition_amd"
    mods = {m: importlib.import_module(f"{pkg}.{m}") for m in ("pipeline", "engine", "session", "app", "video", "onnx_reader", "lower")}
    mods["helpers"] = importlib.import_module("utils.helpers")
    seen = 0
    for mod, attr in set(re.findall(r"\b(pipeline|engine|session|app|video|onnx_reader|lower|helpers)\.([A-Za-z_][A-Za-z0-9_]*)", doc)):
        if attr in ("py", "hip"):
            continue
        # (`app` is also the document's FaceAnalysis instance, `session.run` the reference's ORT call it replaces)
        ok = hasattr(mods[mod], attr) or (mod == "app" and (hasattr(mods["app"].FaceAnalysis, attr) or f"self.{attr} =" in inspect.getsource(mods["app"].FaceAnalysis))) or (mod == "session" and hasattr(mods["session"].HipSession, attr))
        assert ok, f"INTEGRATION.md names {mod}.{attr}"
        seen += 1
    owner = {"SCRFD": "models", "ArcFace": "models", "Communicator": "pipeline", "FacePipeline": "pipeline", "GroupedFacePipeline": "pipeline", "FaceAnalysis": "app",
             "HipSession": "session", "GateConfig": "app"}
    for cls, meth in set(re.findall(r"\b(" + "|".join(owner) + r")\.([a-z_][A-Za-z0-9_]*)", doc)):
        m = importlib.import_module(f"{pkg}.{owner[cls]}")
        assert hasattr(getattr(m, cls), meth), f"INTEGRATION.md names {cls}.{meth}"
        seen += 1
    assert seen >= 12
    # the calls the document shows
    from scrfd_arcface_facerecognition_amd.pipeline import Communicator, run_step_distributed
    assert list(inspect.signature(Communicator.__init__).parameters)[1:] == ["ctx", "world", "rank", "exchange_id"]
    assert re.search(r"Communicator\(ctx, world_size, rank, exchange_id\)", doc) and "Communicator.unique_id" not in doc
    p = list(inspect.signature(run_step_distributed).parameters)
    assert p[:9] == ["pipe", "frames_dev", "H", "W", "gallery", "thresh", "q_local", "q_all", "dist"] and {"idx_all", "score_all", "match_scope", "flush"} <= set(p)
    from scrfd_arcface_facerecognition_amd.pipeline import GroupedFacePipeline
    assert "group" in inspect.signature(GroupedFacePipeline.__init__).parameters

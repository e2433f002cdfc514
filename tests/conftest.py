import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def dense_heads(pos, pos_score, pos_bbox, pos_kps, size=640):
    """Rebuild the 9 dense head tensors from the sparse fixture form (tools/gen_golden.py:
    constant 0.125 background score, zero bbox/kps everywhere else)."""
    ns = [(size // s) * (size // s) * 2 for s in (8, 16, 32)]
    total = sum(ns)
    scores = np.full(total, 0.125, dtype=np.float32)
    bbox = np.zeros((total, 4), dtype=np.float32)
    kps = np.zeros((total, 10), dtype=np.float32)
    scores[pos], bbox[pos], kps[pos] = pos_score, pos_bbox, pos_kps
    outs, o = [], 0
    for n in ns:
        outs.append(scores[o:o + n].reshape(n, 1))
        o += n
    o = 0
    for n in ns:
        outs.append(bbox[o:o + n])
        o += n
    o = 0
    for n in ns:
        outs.append(kps[o:o + n])
        o += n
    return outs


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def lib():
    """The ctypes-loaded HIP library; GPU tests go through the C-ABI only."""
    from scrfd_arcface_facerecognition_amd import _lib
    return _lib.load()

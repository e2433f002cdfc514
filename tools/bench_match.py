#!/usr/bin/env python3
"""Gallery-match throughput (fid_match = MFMA cosine GEMM + fused arg-max) at the gallery sizes of
BASELINE.json configs 2-4, with a check that planted matches are found."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scrfd_arcface_facerecognition_amd._lib import Context, check  # noqa: E402
from scrfd_arcface_facerecognition_amd.engine import Gallery  # noqa: E402

ctx = Context(0)
rng = np.random.default_rng(0)
CASES = ((1000, 64), (100_000, 512), (1_000_000, 512), (1_000_000, 10_000))
if len(sys.argv) > 1:                                   # bench_match.py G n: one case (ablation runs)
    CASES = ((int(sys.argv[1]), int(sys.argv[2])),)
for G, n in CASES:
    gal_h = rng.standard_normal((G, 512), dtype=np.float32)
    gal = Gallery(ctx, gal_h)
    emb = rng.standard_normal((n, 512), dtype=np.float32)
    pick = rng.integers(0, G, n)
    emb[::2] = gal_h[pick[::2]] + 0.5 * rng.standard_normal((len(emb[::2]), 512), dtype=np.float32)
    e = ctx.to_device(emb)
    q = ctx.empty((n, 512), np.float16)
    check(ctx.lib.fid_l2_normalize_f16(ctx.handle, C.c_void_p(e.ptr), n, 512, C.c_void_p(q.ptr)))
    idx, sc = ctx.empty((n,), np.int32), ctx.empty((n,), np.float32)
    gal.match_device(q, n, 0.4, idx, sc)
    ctx.sync()
    reps = 5
    ctx.event_record(0)
    for _ in range(reps):
        gal.match_device(q, n, 0.4, idx, sc)
    ctx.event_record(1)
    ms = ctx.elapsed_ms(0, 1) / reps
    ok = (idx.download()[::2] == pick[::2]).mean()
    flops = 2.0 * 512 * G * n
    print(f"G={G:>9,d} n={n:>6,d}: {ms:9.3f} ms  {flops / ms / 1e9:8.1f} TFLOP/s  {n / ms * 1e3:12,.0f} faces/s  "
          f"planted matches found {ok * 100:.1f}%", flush=True)
    gal.close()
    del gal_h

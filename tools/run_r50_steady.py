#!/usr/bin/env python3
"""Steady-state IResNet-50 (or another net) for profiling: tune once, then N plain runs at a batch size (no per-op events),
so that rocprofv3 kernel-trace / PMC sums are dominated by the steady state.  Usage: python tools/run_r50_steady.py [arch] [batch] [runs]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scrfd_arcface_facerecognition_amd import archs  # noqa: E402
from scrfd_arcface_facerecognition_amd._lib import Context  # noqa: E402
from scrfd_arcface_facerecognition_amd.engine import CompiledNet  # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else "arcface_r50"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 500
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 40
ctx = Context(0)
net = archs.ARCHS[arch]()
cn = CompiledNet(ctx, net, archs.synth_params(net, 0), max_batch=batch)
H, W = net.in_hw
imgs = ctx.to_device(np.random.default_rng(0).integers(0, 256, (batch, H, W, 3), dtype=np.uint8))
cn.run_device(imgs, batch)
ctx.sync()
t0 = time.perf_counter()
for _ in range(runs):
    cn.run_device(imgs, batch)
ctx.sync()
dt = (time.perf_counter() - t0) / runs
gf = 2.0 * cn.macs_per_image() * batch / 1e9
print(f"{arch} batch {batch}: {dt * 1e3:.3f} ms per run, {gf / dt / 1e3:.1f} TFLOP/s algorithmic")

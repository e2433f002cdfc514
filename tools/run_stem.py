#!/usr/bin/env python3
"""The SCRFD deep stem alone (conv/s2 - conv - conv - maxpool as ONE fused op) at batch 64 of 640x640 frames, N plain runs: for rocprofv3
kernel-trace / PMC passes of the stem kernel (tools/pmc_mfma.sh stem tools/run_stem.py).  Usage: python tools/run_stem.py [batch] [runs]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scrfd_arcface_facerecognition_amd import archs  # noqa: E402
from scrfd_arcface_facerecognition_amd._lib import Context  # noqa: E402
from scrfd_arcface_facerecognition_amd.archs import Conv, MaxPool, Net  # noqa: E402
from scrfd_arcface_facerecognition_amd.engine import CompiledNet  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 40
net = Net("stem", (640, 640), 127.5, 1.0 / 128.0)
net.add(Conv("stem.0", "input", 3, 28, stride=2, act="relu"))
net.add(Conv("stem.1", "stem.0", 28, 28, act="relu"))
net.add(Conv("stem.2", "stem.1", 28, 56, act="relu"))
net.add(MaxPool("stem.pool", "stem.2", 56))
net.outputs = ["stem.pool"]
ctx = Context(0)
cn = CompiledNet(ctx, net, archs.synth_params(net, 0), max_batch=batch)
assert cn.low.op_names == ["stem.fused"]
imgs = ctx.to_device(np.random.default_rng(0).integers(0, 256, (batch, 640, 640, 3), dtype=np.uint8))
cn.run_device(imgs, batch)
ctx.sync()
t0 = time.perf_counter()
for _ in range(runs):
    cn.run_device(imgs, batch)
ctx.sync()
dt = (time.perf_counter() - t0) / runs
print(f"stem batch {batch}: {dt * 1e6:.1f} us per run, {2.0 * cn.macs_per_image() * batch / dt / 1e12:.1f} TFLOP/s algorithmic")

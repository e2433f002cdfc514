#!/usr/bin/env python3
"""Run one net once at a batch size (diagnostic builds: stamps, ablations).  Usage: python tools/run_net_once.py arch batch"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scrfd_arcface_facerecognition_amd import archs  # noqa: E402
from scrfd_arcface_facerecognition_amd._lib import Context  # noqa: E402
from scrfd_arcface_facerecognition_amd.engine import CompiledNet  # noqa: E402

arch, batch = sys.argv[1], int(sys.argv[2])
ctx = Context(0)
net = archs.ARCHS[arch]()
cn = CompiledNet(ctx, net, archs.synth_params(net, 0), max_batch=batch)
H, W = net.in_hw
imgs = ctx.to_device(np.random.default_rng(0).integers(0, 256, (batch, H, W, 3), dtype=np.uint8))
for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 1):
    cn.run_device(imgs, batch)
ctx.sync()
print("done")

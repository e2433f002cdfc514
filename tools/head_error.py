#!/usr/bin/env python3
"""Max |device - fp32 oracle| of the SCRFD-10G head tensors at full size (640x640, batch 64) for the current kernel plan:
prints per head (score, bbox, kps) maxima over the checked frames.  Usage: [FID_* env] python tools/head_error.py [frames...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import align as oalign, nets as onets  # noqa: E402
from scrfd_arcface_facerecognition_amd import archs  # noqa: E402
from scrfd_arcface_facerecognition_amd._lib import Context  # noqa: E402
from scrfd_arcface_facerecognition_amd.engine import CompiledNet  # noqa: E402
from scrfd_arcface_facerecognition_amd.pipeline import calibrate_detector_bias  # noqa: E402

ctx = Context(0)
B = 64
frames = np.random.default_rng(77).integers(0, 256, (B, 640, 640, 3), dtype=np.uint8)
net = archs.scrfd_10g((640, 640))
P, _ = calibrate_detector_bias(ctx, net, archs.synth_params(net, 0), frames[:8], target=48)
cn = CompiledNet(ctx, net, P, max_batch=B)
cn.run(frames)
idx = [int(a) for a in sys.argv[1:]] or [0, 21, 42, 63]
worst = {}
for fi in idx:
    ref = onets.run_net(net, P, oalign.blob_from_images([frames[fi]], net.in_scale, net.in_mean))
    for name in net.outputs:
        fused = cn.read(name, B)[fi:fi + 1]
        sc, bb, kp = ref[name]
        e = (float(np.abs(fused[..., 0:2].reshape(1, -1, 1) - sc).max()), float(np.abs(fused[..., 2:10].reshape(1, -1, 4) - bb).max()),
             float(np.abs(fused[..., 10:30].reshape(1, -1, 10) - kp).max()))
        worst[name] = tuple(max(a, b) for a, b in zip(worst.get(name, (0, 0, 0)), e))
for k, v in worst.items():
    print(f"{k:14s} score {v[0]:.5f}  bbox {v[1]:.4f}  kps {v[2]:.4f}")

#!/usr/bin/env python3
"""Single-frame latency of the mirrored API (what reference main.py's frame_processor does per frame):
detect(frame) + recognizer(frame, kps) + match against a small gallery."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from models import SCRFD, ArcFace  # noqa: E402
from scrfd_arcface_facerecognition_amd.pipeline import calibrate_detector_bias  # noqa: E402
from scrfd_arcface_facerecognition_amd.session import HipSession  # noqa: E402
from utils.helpers import match_gallery  # noqa: E402

det = SCRFD("synthetic:scrfd_10g", input_size=(640, 640), max_batch=1)
rng = np.random.default_rng(0)
frame = rng.integers(0, 256, (640, 640, 3), dtype=np.uint8)
P, _ = calibrate_detector_bias(det.ctx, det.session.net, det.session.params, frame[None], target=40, max_batch=1)
det.session = HipSession(None, ctx=det.ctx, net=det.session.net, params=P, max_batch=1)
rec = ArcFace("synthetic:arcface_r50", max_batch=1)
gal = rng.standard_normal((5, 512)).astype(np.float32)


def once():
    d, k = det.detect(frame, max_num=1)
    e = rec(frame, k[0])
    return match_gallery(e[None], gal, 0.4)


for _ in range(5):
    once()
ts = []
for _ in range(30):
    t0 = time.perf_counter(); once(); ts.append(time.perf_counter() - t0)
t_det = []
for _ in range(30):
    t0 = time.perf_counter(); det.detect(frame, max_num=1); t_det.append(time.perf_counter() - t0)
print(f"single frame 640x640: detect+align+embed+match median {np.median(ts) * 1e3:.2f} ms (min {min(ts) * 1e3:.2f}); "
      f"detect alone {np.median(t_det) * 1e3:.2f} ms")

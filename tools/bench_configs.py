#!/usr/bin/env python3
"""Side measurements for the other configurations of BASELINE.json (NOT bench lines; the bench line is bench.py):

  cfg1  SCRFD-500M + MobileFaceNet, 1 frame 640x640, 5-entry gallery        (the reference's CPU-runnable case)
  cfg2  SCRFD-10G + IResNet-50, 64 frames, 1 k gallery, F = 8 faces per frame (bench.py quotes F = 1)
  cfg4  IResNet-50 alone: 10 000 resident 112x112 crops in chunks of 500, each chunk matched against a 1 M gallery
  cfg5  SCRFD-2.5G + MobileFaceNet, 32 frames of 1080x1920 (one rank's share), letterboxed on the device, 1 k gallery

Usage: python tools/bench_configs.py [cfg1|cfg2f8|cfg4|cfg5 ...]     -> one JSON object per configuration
Synthetic frames and random-init weights (seed 0), frames resident in HBM before the timed region."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scrfd_arcface_facerecognition_amd import archs  # noqa: E402
from scrfd_arcface_facerecognition_amd._lib import Context  # noqa: E402
from scrfd_arcface_facerecognition_amd.engine import CompiledNet, Gallery  # noqa: E402
from scrfd_arcface_facerecognition_amd.pipeline import FacePipeline, calibrate_detector_bias  # noqa: E402

CFG = {
    "cfg1": dict(det="scrfd_500m", rec="arcface_mbf", B=1, HW=(640, 640), F=1, G=5, steps=200),
    "cfg2f8": dict(det="scrfd_10g", rec="arcface_r50", B=64, HW=(640, 640), F=8, G=1000, steps=20),
    "cfg5": dict(det="scrfd_2.5g", rec="arcface_mbf", B=32, HW=(1080, 1920), F=1, G=1000, steps=30),
}


def run_cfg4(n_crops=10_000, chunk=500, G=1_000_000):
    """BASELINE.json configs[3]: no detector; embed + L2-normalise + match, crops and gallery resident in HBM."""
    import ctypes as C
    from scrfd_arcface_facerecognition_amd._lib import check
    ctx = Context(0)
    net = archs.ARCHS["arcface_r50"]()
    rec = CompiledNet(ctx, net, archs.synth_params(net, 0), max_batch=chunk)
    rng = np.random.default_rng(99)
    gal_h = rng.standard_normal((G, 512), dtype=np.float32)
    gal = Gallery(ctx, gal_h)
    del gal_h
    crops = ctx.to_device(np.random.default_rng(7).integers(0, 256, (n_crops, 112, 112, 3), dtype=np.uint8))
    q = ctx.empty((chunk, 512), np.float16)
    idx, sc = ctx.empty((n_crops,), np.int32), ctx.empty((n_crops,), np.float32)
    emb_ptr, _, _ = rec.tensor(rec.low.outputs[0])
    per = 112 * 112 * 3

    def one_pass():
        for first in range(0, n_crops, chunk):
            nb = min(chunk, n_crops - first)
            check(ctx.lib.fid_net_run(ctx.handle, rec.handle, C.c_void_p(crops.ptr + first * per), nb))
            check(ctx.lib.fid_l2_normalize_f16(ctx.handle, C.c_void_p(emb_ptr), nb, 512, C.c_void_p(q.ptr)))
            check(ctx.lib.fid_match(ctx.handle, gal.handle, C.c_void_p(q.ptr), nb, C.c_float(0.4),
                                    C.c_void_p(idx.ptr + first * 4), C.c_void_p(sc.ptr + first * 4)))

    one_pass()
    ctx.sync()
    t0 = time.perf_counter()
    one_pass()
    ctx.sync()
    dt = time.perf_counter() - t0
    gflop_face = 2.0 * rec.macs_per_image() / 1e9
    print(json.dumps({"config": "cfg4", "rec": "arcface_r50", "crops": n_crops, "chunk": chunk, "gallery": G,
                      "s_per_pass": round(dt, 4), "faces_per_s": round(n_crops / dt, 1),
                      "embed_gflop_per_face": round(gflop_face, 2), "match_gflop_per_face": round(2.0 * 512 * G / 1e9, 3),
                      "tflops_embed_plus_match": round(n_crops * (gflop_face + 2.0 * 512 * G / 1e9) / dt / 1e3, 1)}), flush=True)


def run(name):
    if name == "cfg4":
        return run_cfg4()
    c = CFG[name]
    ctx = Context(0)
    det_net = archs.ARCHS[c["det"]]()
    rec_net = archs.ARCHS[c["rec"]]()
    calib = np.random.default_rng(1234).integers(0, 256, (8, 640, 640, 3), dtype=np.uint8)
    det_P, _ = calibrate_detector_bias(ctx, det_net, archs.synth_params(det_net, 0), calib, target=48)
    rec_P = archs.synth_params(rec_net, 0)
    B, F, (H, W) = c["B"], c["F"], c["HW"]
    det = CompiledNet(ctx, det_net, det_P, max_batch=B)
    rec = CompiledNet(ctx, rec_net, rec_P, max_batch=B * F)
    gal = Gallery(ctx, np.random.default_rng(99).standard_normal((c["G"], 512)).astype(np.float32))
    pipe = FacePipeline(ctx, det, rec, batch=B, faces_per_frame=F)
    frames = ctx.to_device(np.random.default_rng(7).integers(0, 256, (B, H, W, 3), dtype=np.uint8))
    for _ in range(3):
        pipe.run_step(frames, H, W, gal, 0.4)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(c["steps"]):
        pipe.run_step(frames, H, W, gal, 0.4)
    ctx.sync()
    dt = (time.perf_counter() - t0) / c["steps"]
    pipe.post.check()
    faces = int(np.minimum(pipe.post.counts.download(), F).sum())
    print(json.dumps({"config": name, "det": c["det"], "rec": c["rec"], "frames": B, "frame_hw": [H, W], "faces_per_frame_cap": F,
                      "gallery": c["G"], "faces_per_step": faces, "ms_per_step": round(dt * 1e3, 4),
                      "frames_per_s": round(B / dt, 1), "faces_per_s": round(faces / dt, 1)}), flush=True)


if __name__ == "__main__":
    for n in (sys.argv[1:] or list(CFG) + ["cfg4"]):
        run(n)

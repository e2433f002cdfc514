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


def run_cfg4(n_crops=10_000, chunk=500, G=1_000_000, lanes=2):
    """BASELINE.json configs[3]: no detector; embed + L2-normalise + match, crops and gallery resident in HBM.
    Like bench.py, `lanes` independent library contexts (HIP streams) work on alternate chunks, so the tails and the
    bandwidth-bound first layers of one chunk overlap the matrix-bound layers of another; every chunk is still one full
    embed -> normalise -> match pass."""
    import ctypes as C
    from scrfd_arcface_facerecognition_amd._lib import check
    ctxs = [Context(0) for _ in range(lanes)]
    ctx = ctxs[0]
    net = archs.ARCHS["arcface_r50"]()
    P = archs.synth_params(net, 0)
    recs = [CompiledNet(c, net, P, max_batch=chunk) for c in ctxs]
    rng = np.random.default_rng(99)
    gal_h = rng.standard_normal((G, 512), dtype=np.float32)
    gal = Gallery(ctx, gal_h)                      # read-only: shared by all lanes
    del gal_h
    crops = ctx.to_device(np.random.default_rng(7).integers(0, 256, (n_crops, 112, 112, 3), dtype=np.uint8))
    qs = [c.empty((chunk, 512), np.float16) for c in ctxs]
    idx, sc = ctx.empty((n_crops,), np.int32), ctx.empty((n_crops,), np.float32)
    embs = [r.tensor(r.low.outputs[0])[0] for r in recs]
    per = 112 * 112 * 3

    def one_pass():
        for k, first in enumerate(range(0, n_crops, chunk)):
            ln = k % lanes
            c, rec = ctxs[ln], recs[ln]
            nb = min(chunk, n_crops - first)
            check(c.lib.fid_net_run(c.handle, rec.handle, C.c_void_p(crops.ptr + first * per), nb))
            check(c.lib.fid_l2_normalize_f16(c.handle, C.c_void_p(embs[ln]), nb, 512, C.c_void_p(qs[ln].ptr)))
            check(c.lib.fid_match(c.handle, gal.handle, C.c_void_p(qs[ln].ptr), nb, C.c_float(0.4),
                                  C.c_void_p(idx.ptr + first * 4), C.c_void_p(sc.ptr + first * 4)))

    def sync():
        for c in ctxs:
            c.sync()

    for ln in range(lanes):                        # every lane tunes its kernels alone
        check(ctxs[ln].lib.fid_net_run(ctxs[ln].handle, recs[ln].handle, C.c_void_p(crops.ptr), chunk))
        ctxs[ln].sync()
    one_pass()
    sync()
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        one_pass()
        sync()
        times.append(time.perf_counter() - t0)
    dt = float(np.median(times))
    gflop_face = 2.0 * recs[0].macs_per_image() / 1e9
    print(json.dumps({"config": "cfg4", "rec": "arcface_r50", "crops": n_crops, "chunk": chunk, "gallery": G, "lanes": lanes,
                      "s_per_pass": round(dt, 4), "s_per_pass_runs": [round(t, 4) for t in times], "faces_per_s": round(n_crops / dt, 1),
                      "embed_gflop_per_face": round(gflop_face, 2), "match_gflop_per_face": round(2.0 * 512 * G / 1e9, 3),
                      "tflops_embed_plus_match": round(n_crops * (gflop_face + 2.0 * 512 * G / 1e9) / dt / 1e3, 1),
                      "tflops_embed_only_share": round(n_crops * gflop_face / dt / 1e3, 1)}), flush=True)


def run(name):
    if name == "cfg4":
        return run_cfg4()
    if name == "cfg4x1":
        return run_cfg4(lanes=1)
    if name == "cfg4c585":                          # chunks of 585 crops = 512 STRIP tiles of 14 x 16 on IResNet's 14x14 stage: two full rounds of pair items on 256 CUs
        return run_cfg4(chunk=585)
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
    for n in (sys.argv[1:] or list(CFG) + ["cfg4", "cfg4c585"]):
        run(n)

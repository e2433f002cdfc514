#!/usr/bin/env python3
"""Where does the stride-32 head of SCRFD-10G lose precision at full size?  

Runs the batch-64 detector of tests/test_gpu_fullsize_properties.py with EVERY tensor kept (own slot each), then for one frame
prints per op
  cum   max |device - fp32 oracle chain| / max |oracle|          (error accumulated from the frame down to this tensor)
  local max |device - fp32 conv of the DEVICE's own input| / max  (what this op alone adds: fp16 operands, fp32 accumulate, fp16 store)
so a layer that adds more than its fp16 rounding shows up in `local`, and plain depth shows up as `cum` growing with small `local`s.
Usage: [FID_* env] python tools/head_trace.py [frame = 63]"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import align as oalign, nets as onets  # noqa: E402
from scrfd_arcface_facerecognition_amd import archs  # noqa: E402
from scrfd_arcface_facerecognition_amd._lib import Context  # noqa: E402
from scrfd_arcface_facerecognition_amd.engine import CompiledNet  # noqa: E402
from scrfd_arcface_facerecognition_amd.pipeline import calibrate_detector_bias  # noqa: E402

fi = int(sys.argv[1]) if len(sys.argv) > 1 else 63
ctx = Context(0)
B = 64
frames = np.random.default_rng(77).integers(0, 256, (B, 640, 640, 3), dtype=np.uint8)
net = archs.scrfd_10g((640, 640))
P, _ = calibrate_detector_bias(ctx, net, archs.synth_params(net, 0), frames[:8], target=48)
heads = list(net.outputs)
stem = {"stem.0", "stem.1", "stem.2"}
net.outputs = [n.name for n in net.nodes if n.name not in stem]          # every tensor kept; the stem stays one fused op
cn = CompiledNet(ctx, net, P, max_batch=B)
cn.run(frames)
blob = oalign.blob_from_images([frames[fi]], net.in_scale, net.in_mean)
ref = onets.run_net(net, P, blob, keep=tuple(net.outputs))


def dev(name):
    """device tensor of frame fi as float32 NCHW (heads: [1, H, W, 30])"""
    t = cn.read(name, B)[fi:fi + 1]
    return t if name in heads else np.transpose(t, (0, 3, 1, 2))


def bn(x, prefix):
    a = P[prefix + ".gamma"].astype(np.float64) / np.sqrt(P[prefix + ".var"].astype(np.float64) + archs.BN_EPS)
    b = P[prefix + ".beta"].astype(np.float64) - P[prefix + ".mean"].astype(np.float64) * a
    return x * torch.from_numpy(a.astype(np.float32)).view(1, -1, 1, 1) + torch.from_numpy(b.astype(np.float32)).view(1, -1, 1, 1)


print(f"frame {fi}; picks: " + ", ".join(sorted({f"gen{p['gen']}" for p in cn.plans()})))
print(f"{'op':20s} {'cum':>9s} {'local':>9s}   max|ref|")
for n in net.nodes:
    if n.name in stem:
        continue
    if n.kind == "dethead":
        sc, bb, kp = ref[n.name]
        d = dev(n.name)
        x = torch.from_numpy(dev(n.src))
        A = n.num_anchors
        s = float(P[n.wname + ".bbox.scale"][0])
        lsc = torch.sigmoid(F.conv2d(x, torch.from_numpy(P[n.wname + ".cls.weight"]), torch.from_numpy(P[n.wname + ".cls.bias"]), 1, 1))
        lsc = lsc.permute(0, 2, 3, 1).reshape(1, -1, 1).numpy()
        cum = float(np.abs(d[..., 0:A].reshape(1, -1, 1) - sc).max())
        loc = float(np.abs(d[..., 0:A].reshape(1, -1, 1) - lsc).max())
        print(f"{n.name:20s} {cum:9.5f} {loc:9.5f}   (sigmoid scores, absolute)")
        continue
    r = ref[n.name]
    d = dev(n.name)
    scale = float(np.abs(r).max()) + 1e-9
    cum = float(np.abs(d - r).max()) / scale
    loc = float("nan")
    if n.kind == "conv" and n.src != "input" and n.src not in stem:
        x = torch.from_numpy(dev(n.src))
        if n.pre_bn:
            x = bn(x, n.wname + ".pre_bn")
        if n.pre_avgpool:
            x = F.avg_pool2d(x, 2, 2)
        y = F.conv2d(x, torch.from_numpy(P[n.wname + ".weight"]), torch.from_numpy(P[n.wname + ".bias"]) if n.bias else None, n.stride, n.pad, 1, n.groups)
        if n.post_bn:
            y = bn(y, n.wname + ".post_bn")
        if n.res is not None:
            rr = torch.from_numpy(dev(n.res))
            if n.res_up2:
                rr = F.interpolate(rr, scale_factor=2, mode="nearest")
            y = y + rr
        if n.act == "relu":
            y = F.relu(y)
        elif n.act == "prelu":
            y = F.prelu(y, torch.from_numpy(P[n.wname + ".prelu"]))
        loc = float(np.abs(d - y.numpy()).max()) / scale
    print(f"{n.name:20s} {cum:9.5f} {loc:9.5f}   {scale:8.3f}")

#!/usr/bin/env python3
"""Per-layer device time of a net on the GPU (HIP events around every launch): which layers are
far from the MFMA roofline.  Usage: python tools/profile_ops.py [arch] [batch]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import node_macs  # noqa: E402
from scrfd_arcface_facerecognition_amd import archs  # noqa: E402
from scrfd_arcface_facerecognition_amd._lib import Context  # noqa: E402
from scrfd_arcface_facerecognition_amd.engine import CompiledNet  # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else "arcface_r50"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ctx = Context(0)
net = archs.ARCHS[arch]()
cn = CompiledNet(ctx, net, archs.synth_params(net, 0), max_batch=batch)
H, W = net.in_hw
imgs = ctx.to_device(np.random.default_rng(0).integers(0, 256, (batch, H, W, 3), dtype=np.uint8))
best = None
for _ in range(5):
    ms = cn.run_profiled(imgs, batch)
    best = ms if best is None else np.minimum(best, ms)
shp = archs.infer_shapes(net)
by_name = {nd.name: nd for nd in net.nodes}
print(f"{arch} batch {batch}: total {best.sum():.3f} ms")
print(f"{'op':28s} {'type':4s} {'out CxHxW':>14s} {'k':>2s} {'s':>2s} {'GFLOP':>8s} {'us':>8s} {'TFLOP/s':>8s}")
for oi, names in enumerate(cn.low.op_nodes):
    t = int(cn.low.ops[oi, 0])
    fl = 2.0 * sum(node_macs(net, by_name[nm]) for nm in names) * batch
    node = by_name[names[-1]]
    c, h, w = shp[node.name]
    k = getattr(node, "k", 0)
    s = getattr(node, "stride", 0)
    print(f"{cn.low.op_names[oi]:28s} {t:4d} {f'{c}x{h}x{w}':>14s} {k:2d} {s:2d} {fl / 1e9:8.2f} {best[oi] * 1e3:8.1f} "
          f"{fl / best[oi] / 1e9 if best[oi] > 0 else 0:8.1f}")

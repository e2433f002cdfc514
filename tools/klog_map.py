#!/usr/bin/env python3
"""Algorithmic GFLOP per KERNEL NAME of one run of a net (`executed_over_algorithmic` per kernel).

Runs the net twice in a child process with FID_KLOG=1 (csrc/ctx.hip prints the kernel every launcher is about to launch, csrc/net.hip the op it
belongs to), takes the second run's op -> kernel map (the first one may tune), joins it with every op's algorithmic FLOPs (2 x MACs of the true
channel counts, bench.node_macs) and prints JSON {kernel name as tools/pmc_mfma.py normalises it: GFLOP per run}.  An op the consumer absorbed launches
nothing: its FLOPs go to the absorbing op's kernel (they are computed there).  Usage: python tools/klog_map.py arch batch > map.json"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from scrfd_arcface_facerecognition_amd import archs
from scrfd_arcface_facerecognition_amd._lib import Context
from scrfd_arcface_facerecognition_amd.engine import CompiledNet
arch, batch = sys.argv[1], int(sys.argv[2])
ctx = Context(0)
net = archs.ARCHS[arch]()
cn = CompiledNet(ctx, net, archs.synth_params(net, 0), max_batch=batch)
H, W = net.in_hw
imgs = ctx.to_device(np.random.default_rng(0).integers(0, 256, (batch, H, W, 3), dtype=np.uint8))
for i in range(2):
    sys.stderr.write("[klog] run %%d\n" %% i); sys.stderr.flush()
    cn.run_device(imgs, batch)
    ctx.sync()
""" % ROOT


def normalise(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").replace("fid::", "").split("(")[0]


def main():
    arch, batch = sys.argv[1], int(sys.argv[2])
    env = dict(os.environ, FID_KLOG="1")
    p = subprocess.run([sys.executable, "-c", CHILD, arch, str(batch)], env=env, capture_output=True, text=True)
    if p.returncode != 0:
        sys.exit(p.stderr[-3000:])
    run, op, op_kernels = -1, None, {}
    for line in p.stderr.splitlines():
        m = re.match(r"\[klog\] (run|op|kernel) (.*)", line)
        if not m:
            continue
        if m.group(1) == "run":
            run = int(m.group(2))
        elif m.group(1) == "op":
            op = int(m.group(2))
        elif run == 1 and op is not None:
            op_kernels.setdefault(op, []).append(m.group(2).strip())
    names = sorted({k for v in op_kernels.values() for k in v})
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines() if names else []
    pretty = dict(zip(names, (normalise(d) for d in dem)))
    from bench import node_macs
    from scrfd_arcface_facerecognition_amd import archs
    from scrfd_arcface_facerecognition_amd.lower import lower
    net = archs.ARCHS[arch]()
    low = lower(net, archs.synth_params(net, 0))
    by_name = {nd.name: nd for nd in net.nodes}
    out, unmapped, carry = {}, 0.0, 0.0
    for oi, nodes in enumerate(low.op_nodes):
        gf = 2.0 * sum(node_macs(net, by_name[nm]) for nm in nodes) * batch / 1e9
        ks = op_kernels.get(oi)
        if not ks:
            if int(low.ops[oi, 0]) == 2 and int(low.ops[oi, 23]) == 0 and int(low.ops[oi, 29]) > 0:
                carry += gf                                  # a shortcut conv its consumer absorbed: computed by the consumer's kernel
            else:
                unmapped += gf
            continue
        k = pretty[ks[0]]                                    # (an op's first kernel is its conv; split-K epilogues / repacks follow)
        out[k] = out.get(k, 0.0) + gf
    if carry:
        # absorbed shortcuts ride in the stride-2 convs (generation 12): spread over the kernels of the ops that name a second weight image
        hosts = [pretty[op_kernels[oi][0]] for oi in op_kernels if int(low.ops[oi, 0]) == 2 and int(low.ops[oi, 23]) > 0]
        for h in hosts:
            out[h] = out.get(h, 0.0) + carry / len(hosts)
    print(json.dumps({"arch": arch, "batch": batch, "gflop_per_run_by_kernel": {k: round(v, 3) for k, v in sorted(out.items())},
                      "gflop_unmapped": round(unmapped, 3)}, indent=1))


if __name__ == "__main__":
    main()

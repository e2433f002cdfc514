#!/usr/bin/env python3
"""Write an archs.Net + parameters as an ONNX file (hand-rolled protobuf, no `onnx` package).
Test infrastructure for scrfd_arcface_facerecognition_amd/onnx_reader.py: it produces files with the node
vocabulary of the five upstream models (Conv, BatchNormalization, Relu, PRelu, Add, MaxPool, AveragePool,
Resize, Sigmoid, Mul, Transpose, Reshape, Flatten, Gemm), either with explicit BatchNormalization nodes
(`fold_bn=False`, like w600k_r50.onnx) or with them folded into the convolutions (`fold_bn=True`, like a
simplified det_10g.onnx)."""
import struct
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from scrfd_arcface_facerecognition_amd.archs import BN_EPS  # noqa: E402


def _v(x):
    x &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = x & 0x7F
        x >>= 7
        out.append(b | (0x80 if x else 0))
        if not x:
            return bytes(out)


def _ld(f, payload):
    return _v((f << 3) | 2) + _v(len(payload)) + payload


def _vi(f, x):
    return _v((f << 3) | 0) + _v(x)


def _str(f, s):
    return _ld(f, s.encode())


def tensor(name, arr):
    arr = np.ascontiguousarray(arr)
    dt = {np.dtype(np.float32): 1, np.dtype(np.int64): 7}[arr.dtype]
    return b"".join(_vi(1, d) for d in arr.shape) + _vi(2, dt) + _str(8, name) + _ld(9, arr.tobytes())


def attr(name, v):
    b = _str(1, name)
    if isinstance(v, float):
        return b + _v((2 << 3) | 5) + struct.pack("<f", v) + _vi(20, 1)
    if isinstance(v, int):
        return b + _vi(3, v) + _vi(20, 2)
    if isinstance(v, str):
        return b + _ld(4, v.encode()) + _vi(20, 3)
    if isinstance(v, (list, tuple)) and all(isinstance(x, int) for x in v):
        return b + b"".join(_vi(8, x) for x in v) + _vi(20, 7)
    if isinstance(v, (list, tuple)):
        return b + b"".join(_v((7 << 3) | 5) + struct.pack("<f", x) for x in v) + _vi(20, 6)
    raise TypeError(v)


def node(op, inputs, outputs, name="", **attrs):
    b = b"".join(_str(1, i) for i in inputs) + b"".join(_str(2, o) for o in outputs) + _str(3, name) + _str(4, op)
    return b + b"".join(_ld(5, attr(k, v)) for k, v in attrs.items())


def value_info(name, shape):
    dims = b"".join(_ld(1, _vi(1, d) if isinstance(d, int) else _str(2, str(d))) for d in shape)
    ttype = _vi(1, 1) + _ld(2, dims)
    return _str(1, name) + _ld(2, _ld(1, ttype))


def export(net, P, fold_bn=False, batch_dim=1, dynamic_reshape=False, upsample="scales", slope_rank=3):
    """dynamic_reshape: a detector output's Reshape takes its shape from a Shape -> Gather -> Unsqueeze -> Concat chain ([N, -1, C]: the form
    dynamic-batch exports have) instead of a constant [-1, C]; upsample: "scales" (Resize with a scales input), "sizes" (Resize with the
    target size as 4th input) or "op9" (the opset-9 Upsample node); slope_rank: PRelu slopes as [C, 1, 1] (3) or [1, C, 1, 1] (4)."""
    nodes, inits = [], []
    uid = [0]

    def fresh(prefix):
        uid[0] += 1
        return f"{prefix}_{uid[0]}"

    def init(name, arr):
        inits.append(tensor(name, np.asarray(arr)))
        return name

    def bn(x, prefix, c_hint):
        y = fresh("bn")
        names = [init(f"{prefix}.{k}", P[f"{prefix}.{k}"]) for k in ("gamma", "beta", "mean", "var")]
        nodes.append(node("BatchNormalization", [x] + names, [y], name=y, epsilon=float(BN_EPS)))
        return y

    from scrfd_arcface_facerecognition_amd.archs import infer_shapes
    shapes = infer_shapes(net)
    t = {"input": "input.1"}
    H, W = net.in_hw
    outputs = []
    head_out = {"score": [], "bbox": [], "kps": []}
    for n in net.nodes:
        if n.kind == "conv":
            x = t[n.src]
            if n.pre_bn:
                x = bn(x, n.wname + ".pre_bn", n.cin)
            if n.pre_avgpool:
                y = fresh("avgpool")
                nodes.append(node("AveragePool", [x], [y], name=y, kernel_shape=[2, 2], strides=[2, 2]))
                x = y
            Wt = P[n.wname + ".weight"].astype(np.float32)
            b = P[n.wname + ".bias"].astype(np.float32) if n.bias else None
            post = n.post_bn
            if post and fold_bn:
                g, be, m, v = (P[f"{n.wname}.post_bn.{k}"].astype(np.float64) for k in ("gamma", "beta", "mean", "var"))
                a = g / np.sqrt(v + BN_EPS)
                Wt = (Wt.astype(np.float64) * a[:, None, None, None]).astype(np.float32)
                b = ((b.astype(np.float64) if b is not None else 0.0) * a + (be - m * a)).astype(np.float32)
                post = False
            ins = [x, init(fresh(n.name + ".W"), Wt)]
            if b is not None:
                ins.append(init(fresh(n.name + ".B"), b))
            y = fresh("conv")
            nodes.append(node("Conv", ins, [y], name=y, kernel_shape=[n.k, n.k], strides=[n.stride, n.stride],
                              pads=[n.pad] * 4, group=n.groups, dilations=[1, 1]))
            if post:
                y = bn(y, n.wname + ".post_bn", n.cout)
            if n.res is not None:
                r = t[n.res]
                if n.res_up2:
                    u = fresh("resize")
                    sc = init(fresh("scales"), np.array([1, 1, 2, 2], np.float32))
                    if upsample == "op9":
                        nodes.append(node("Upsample", [r, sc], [u], name=u, mode="nearest"))
                    elif upsample == "sizes":
                        hh, ww = shapes[n.name][1], shapes[n.name][2]
                        sz = init(fresh("sizes"), np.array([batch_dim, n.cout, hh, ww], np.int64))
                        nodes.append(node("Resize", [r, "", "", sz], [u], name=u, mode="nearest"))
                    else:
                        nodes.append(node("Resize", [r, "", sc], [u], name=u, mode="nearest"))
                    r = u
                z = fresh("add")
                nodes.append(node("Add", [y, r], [z], name=z))
                y = z
            if n.act == "relu":
                z = fresh("relu")
                nodes.append(node("Relu", [y], [z], name=z))
                y = z
            elif n.act == "prelu":
                z = fresh("prelu")
                nodes.append(node("PRelu", [y, init(fresh(n.name + ".slope"), P[n.wname + ".prelu"].reshape((-1, 1, 1) if slope_rank == 3 else (1, -1, 1, 1)))], [z], name=z))
                y = z
            t[n.name] = y
        elif n.kind == "maxpool":
            y = fresh("maxpool")
            nodes.append(node("MaxPool", [t[n.src]], [y], name=y, kernel_shape=[n.k, n.k], strides=[n.stride, n.stride],
                              pads=[n.pad] * 4))
            t[n.name] = y
        elif n.kind == "fc":
            x = t[n.src]
            if n.pre_bn:
                x = bn(x, n.wname + ".pre_bn", n.c)
            f = fresh("flatten")
            nodes.append(node("Flatten", [x], [f], name=f, axis=1))
            ins = [f, init("fc.W", P[n.wname + ".weight"])]
            if n.bias:
                ins.append(init("fc.B", P[n.wname + ".bias"]))
            y = fresh("gemm")
            nodes.append(node("Gemm", ins, [y], name=y, transB=1))
            if n.post_bn:
                y = bn(y, n.wname + ".post_bn", n.cout)
            t[n.name] = y
        elif n.kind == "dethead":
            x = t[n.src]
            for part, c in (("cls", 1), ("bbox", 4), ("kps", 10)):
                y = fresh(part)
                nodes.append(node("Conv", [x, init(fresh(part + ".W"), P[f"{n.wname}.{part}.weight"]),
                                           init(fresh(part + ".B"), P[f"{n.wname}.{part}.bias"])], [y], name=y,
                                  kernel_shape=[n.k, n.k], strides=[1, 1], pads=[n.k // 2] * 4, group=1, dilations=[1, 1]))
                if part == "cls":
                    z = fresh("sigmoid")
                    nodes.append(node("Sigmoid", [y], [z], name=z))
                    y = z
                if part == "bbox":
                    z = fresh("mul")
                    nodes.append(node("Mul", [y, init(fresh("scale"), P[n.wname + ".bbox.scale"].reshape(()))], [z], name=z))
                    y = z
                z = fresh("transpose")
                nodes.append(node("Transpose", [y], [z], name=z, perm=[0, 2, 3, 1]))
                o = fresh("out_" + part)
                if dynamic_reshape:
                    sh, g, u2, cat = fresh("shape_of"), fresh("gather"), fresh("unsq"), fresh("concat")
                    nodes.append(node("Shape", [z], [sh], name=sh))
                    nodes.append(node("Gather", [sh, init(fresh("idx"), np.array(0, np.int64))], [g], name=g, axis=0))
                    nodes.append(node("Unsqueeze", [g], [u2], name=u2, axes=[0]))
                    nodes.append(node("Concat", [u2, init(fresh("m1"), np.array([-1], np.int64)), init(fresh("cc"), np.array([c], np.int64))], [cat], name=cat, axis=0))
                    nodes.append(node("Reshape", [z, cat], [o], name=o))
                else:
                    nodes.append(node("Reshape", [z, init(fresh("shape"), np.array([-1, c], np.int64))], [o], name=o))
                head_out[{"cls": "score"}.get(part, part)].append(o)
        else:
            raise ValueError(n.kind)
    if head_out["score"]:
        outputs = head_out["score"] + head_out["bbox"] + head_out["kps"]
        out_infos = [value_info(o, ["N", "A", c] if dynamic_reshape else ["N", c]) for o, c in zip(outputs, [1] * 3 + [4] * 3 + [10] * 3)]
    else:
        outputs = [t[net.outputs[0]]]
        out_infos = [value_info(outputs[0], [batch_dim, 512])]
    graph = b"".join(_ld(1, x) for x in nodes) + _str(2, net.name) + b"".join(_ld(5, x) for x in inits)
    graph += _ld(11, value_info("input.1", [batch_dim, 3, H, W])) + b"".join(_ld(12, x) for x in out_infos)
    model = _vi(1, 7) + _str(2, "faceid-export") + _ld(7, graph) + _ld(8, _str(1, "") + _vi(2, 11))
    return model

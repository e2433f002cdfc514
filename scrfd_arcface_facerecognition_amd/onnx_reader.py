"""Dependency-free ONNX ingestion: read the weights AND the graph of a
`.onnx` file with a ~100-line protobuf wire-format parser (no `onnx`, no `onnxruntime`), and convert
the graph structurally into the unfused IR of archs.py, so that
`SCRFD("./weights/det_10g.onnx")` / `ArcFace("./weights/w600k_r50.onnx")` (reference main.py:19-30,
download.sh:12-16) work when a user supplies the files.

The converter follows dataflow, not node order or tensor names:
  Conv [+ BatchNormalization] [+ Add(shortcut | Resize-nearest-2x(x))] [+ Relu | PRelu]   -> archs.Conv
  BatchNormalization feeding a Conv (IResNet bn1)                                          -> Conv.pre_bn
  AveragePool(2,2) feeding a 1x1 Conv ("avg_down")                                          -> Conv.pre_avgpool
  MaxPool                                                                                   -> archs.MaxPool
  [BatchNormalization] Flatten Gemm [BatchNormalization]                                    -> archs.FC
  3 sibling convs whose results leave through Sigmoid/Mul/Transpose/Reshape to graph outputs -> archs.DetHead
BatchNorms already folded into the convs by the exporter simply do not appear (the Conv then has a bias).

STATUS: exercised by a round trip through tools/export_onnx.py (our own writer, both with BatchNorm
nodes and with folded ones) for all five architectures.  No real insightface file exists offline, so
agreement with the upstream exporters' exact node patterns is UNPINNED.
"""
from __future__ import annotations

import struct
from typing import Dict, List

import numpy as np

from .archs import Conv, DetHead, FC, MaxPool, Net

# ------------------------------------------------------------------------------------------------
# protobuf wire format
# ------------------------------------------------------------------------------------------------


def _varint(b: bytes, i: int):
    r = s = 0
    while True:
        c = b[i]
        i += 1
        r |= (c & 0x7F) << s
        if not c & 0x80:
            return r, i
        s += 7


def _fields(b: bytes):
    """yield (field_number, wire_type, value) over one message; length-delimited values as memoryview slices"""
    i, n = 0, len(b)
    while i < n:
        key, i = _varint(b, i)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(b, i)
        elif wt == 1:
            v = b[i:i + 8]; i += 8
        elif wt == 2:
            ln, i = _varint(b, i)
            v = b[i:i + ln]; i += ln
        elif wt == 5:
            v = b[i:i + 4]; i += 4
        else:
            raise ValueError(f"unsupported wire type {wt}")
        yield f, wt, v


def _packed_ints(v, wt):
    if wt == 0:
        return [v]
    out, i = [], 0
    while i < len(v):
        x, i = _varint(v, i)
        out.append(x)
    return out


def _sint(x):          # int64 two's complement of a varint
    return x - (1 << 64) if x >= (1 << 63) else x


def _tensor(b: bytes):
    dims, dtype, name, raw, floats, int64s = [], 1, "", None, [], []
    for f, wt, v in _fields(b):
        if f == 1: dims += [_sint(x) for x in _packed_ints(v, wt)]
        elif f == 2: dtype = v
        elif f == 8: name = bytes(v).decode()
        elif f == 9: raw = bytes(v)
        elif f == 4: floats += list(struct.unpack(f"<{len(v) // 4}f", v)) if wt == 2 else [struct.unpack("<f", v)[0]]
        elif f == 7: int64s += [_sint(x) for x in _packed_ints(v, wt)]
    np_dt = {1: np.float32, 7: np.int64, 10: np.float16, 6: np.int32, 11: np.float64}.get(dtype)
    if np_dt is None:
        raise ValueError(f"tensor {name}: unsupported data_type {dtype}")
    if raw is not None:
        arr = np.frombuffer(raw, dtype=np_dt).copy()
    elif dtype == 7:
        arr = np.asarray(int64s, dtype=np.int64)
    else:
        arr = np.asarray(floats, dtype=np_dt)
    return name, arr.reshape(dims) if dims else arr.reshape(())


def _attribute(b: bytes):
    name, val = "", None
    ints, floats = [], []
    for f, wt, v in _fields(b):
        if f == 1: name = bytes(v).decode()
        elif f == 2: val = struct.unpack("<f", v)[0]
        elif f == 3: val = _sint(v)
        elif f == 4: val = bytes(v)
        elif f == 5: val = _tensor(bytes(v))[1]
        elif f == 7: floats += list(struct.unpack(f"<{len(v) // 4}f", v)) if wt == 2 else [struct.unpack("<f", v)[0]]
        elif f == 8: ints += [_sint(x) for x in _packed_ints(v, wt)]
    if ints: val = ints
    elif floats: val = floats
    return name, val


class OnnxNode:
    def __init__(self):
        self.op, self.name, self.inputs, self.outputs, self.attrs = "", "", [], [], {}


def parse_onnx(data: bytes):
    """-> (nodes, initializers {name: ndarray}, graph input names, graph output names, input shape)"""
    graph = None
    for f, wt, v in _fields(memoryview(data)):
        if f == 7:
            graph = v
    if graph is None:
        raise ValueError("no GraphProto in file")
    nodes, inits, g_in, g_out, in_shape = [], {}, [], [], None
    for f, wt, v in _fields(graph):
        if f == 1:
            n = OnnxNode()
            for f2, wt2, v2 in _fields(v):
                if f2 == 1: n.inputs.append(bytes(v2).decode())
                elif f2 == 2: n.outputs.append(bytes(v2).decode())
                elif f2 == 3: n.name = bytes(v2).decode()
                elif f2 == 4: n.op = bytes(v2).decode()
                elif f2 == 5:
                    k, a = _attribute(bytes(v2))
                    n.attrs[k] = a
            nodes.append(n)
        elif f == 5:
            name, arr = _tensor(bytes(v))
            inits[name] = arr
        elif f in (11, 12):
            name, shape = "", None
            for f2, wt2, v2 in _fields(v):
                if f2 == 1: name = bytes(v2).decode()
                elif f2 == 2:        # TypeProto -> tensor_type(1) -> shape(2) -> dim(1) -> dim_value(1)
                    for f3, _, v3 in _fields(v2):
                        if f3 == 1:
                            for f4, _, v4 in _fields(v3):
                                if f4 == 2:
                                    shape = []
                                    for f5, _, v5 in _fields(v4):
                                        if f5 == 1:
                                            dv = None
                                            for f6, wt6, v6 in _fields(v5):
                                                if f6 == 1: dv = _sint(v6)
                                            shape.append(dv)
            (g_in if f == 11 else g_out).append(name)
            if f == 11 and name not in inits and shape is not None and in_shape is None:
                in_shape = shape
    g_in = [n for n in g_in if n not in inits]
    return nodes, inits, g_in, g_out, in_shape


# ------------------------------------------------------------------------------------------------
# graph -> IR
# ------------------------------------------------------------------------------------------------


def _bn_params(inits, n: OnnxNode):
    g, b, m, v = (np.asarray(inits[x], dtype=np.float32) for x in n.inputs[1:5])
    eps = float(n.attrs.get("epsilon", 1e-5))
    if abs(eps - 1e-5) > 1e-9:           # the IR uses eps = 1e-5: re-express the statistics for that eps
        v = (v + eps - 1e-5).astype(np.float32)
    return {"gamma": g, "beta": b, "mean": m, "var": v}


def onnx_to_ir(data: bytes, name: str, in_hw=None, in_mean=127.5, in_scale=None):
    """Structural conversion.  Returns (archs.Net, params)."""
    nodes, inits, g_in, g_out, in_shape = parse_onnx(data)
    consumers: Dict[str, List[OnnxNode]] = {}
    producer: Dict[str, OnnxNode] = {}
    for n in nodes:
        for o in n.outputs:
            producer[o] = n
        for i in n.inputs:
            consumers.setdefault(i, []).append(n)
    if in_hw is None:
        if not in_shape or len(in_shape) != 4 or not in_shape[2] or not in_shape[3] or in_shape[2] < 0:
            raise ValueError("the ONNX input has a dynamic size: pass in_hw")
        in_hw = (int(in_shape[2]), int(in_shape[3]))
    P: Dict[str, np.ndarray] = {}
    net = Net(name, tuple(in_hw), in_mean, in_scale if in_scale is not None else 1.0 / 128.0)
    ir_of: Dict[str, str] = {g_in[0]: "input"}        # ONNX tensor -> IR tensor that carries the same value
    pending_bn: Dict[str, OnnxNode] = {}               # ONNX tensor = BN(x) waiting for its Conv / Gemm consumer
    pending_pool: Dict[str, str] = {}                  # ONNX tensor = AvgPool2(x): value is x's ONNX tensor
    pending_up: Dict[str, str] = {}                    # ONNX tensor = Resize2x(x)
    conv_of: Dict[str, Conv] = {}                      # IR name -> node (still open for BN / Add / act fusion)
    closed = set()
    head_convs = []
    cnt = {"c": 0}

    def only_consumer(t, op=None):
        c = consumers.get(t, [])
        return len(c) == 1 and (op is None or c[0].op in op)

    for n in nodes:
        op = n.op
        if op == "Conv":
            x = n.inputs[0]
            W = np.asarray(inits[n.inputs[1]], dtype=np.float32)
            cout, cin_g, kh, kw = W.shape
            groups = int(n.attrs.get("group", 1))
            strides = n.attrs.get("strides", [1, 1]); pads = n.attrs.get("pads", [0, 0, 0, 0])
            assert kh == kw and strides[0] == strides[1] and len(set(pads)) == 1, f"{n.name}: anisotropic conv"
            nm = f"conv{cnt['c']}"; cnt["c"] += 1
            pre_bn, pre_pool = None, False
            if x in pending_pool:
                pre_pool, x = True, pending_pool[x]
            if x in pending_bn:
                pre_bn, x = pending_bn[x], pending_bn[x].inputs[0]
            node = Conv(nm, ir_of[x], cin_g * groups, cout, k=kh, stride=strides[0], pad=pads[0], bias=len(n.inputs) > 2,
                        pre_bn=pre_bn is not None, post_bn=False, act="none", pre_avgpool=pre_pool, groups=groups)
            P[nm + ".weight"] = W
            if len(n.inputs) > 2:
                P[nm + ".bias"] = np.asarray(inits[n.inputs[2]], dtype=np.float32)
            if pre_bn is not None:
                for k2, v2 in _bn_params(inits, pre_bn).items():
                    P[f"{nm}.pre_bn.{k2}"] = v2
            net.add(node)
            conv_of[nm] = node
            ir_of[n.outputs[0]] = nm
        elif op == "BatchNormalization":
            x = n.inputs[0]
            src = ir_of.get(x)
            c = conv_of.get(src)
            if c is not None and src not in closed and not c.post_bn and c.res is None and c.act == "none" and only_consumer(x):
                c.post_bn = True
                for k2, v2 in _bn_params(inits, n).items():
                    P[f"{src}.post_bn.{k2}"] = v2
                ir_of[n.outputs[0]] = src
            else:
                pending_bn[n.outputs[0]] = n
        elif op in ("Relu", "PRelu", "LeakyRelu"):
            x = n.inputs[0]
            src = ir_of[x]
            c = conv_of.get(src)
            assert c is not None and src not in closed and c.act == "none" and only_consumer(x), f"{n.name}: activation on a shared tensor"
            if op == "Relu":
                c.act = "relu"
            elif op == "PRelu":
                c.act = "prelu"
                P[src + ".prelu"] = np.asarray(inits[n.inputs[1]], dtype=np.float32).reshape(-1)
            else:
                raise ValueError("LeakyRelu is not part of these nets")
            ir_of[n.outputs[0]] = src
            closed.add(src)
        elif op == "Add":
            a_, b_ = n.inputs
            def open_conv(t):
                s = ir_of.get(t)
                c = conv_of.get(s)
                return c if (c is not None and s not in closed and c.res is None and c.act == "none" and only_consumer(t)) else None
            ca, cb = open_conv(a_), open_conv(b_)
            # the conv that is fused with the add is the one produced LAST (its shortcut already exists)
            pick = None
            if ca is not None and cb is not None:
                pick = (a_, b_) if net.nodes.index(ca) > net.nodes.index(cb) else (b_, a_)
            elif ca is not None:
                pick = (a_, b_)
            elif cb is not None:
                pick = (b_, a_)
            assert pick is not None, f"{n.name}: Add without a fusable conv operand"
            main, other = pick
            c = conv_of[ir_of[main]]
            if other in pending_up:
                c.res, c.res_up2 = ir_of[pending_up[other]], True
            else:
                c.res = ir_of[other]
            # a conv that is ALSO consumed elsewhere must precede; keep order: the res tensor must already exist
            ir_of[n.outputs[0]] = c.name
        elif op == "MaxPool":
            k = n.attrs["kernel_shape"][0]; s = n.attrs.get("strides", [1, 1])[0]; p = n.attrs.get("pads", [0, 0, 0, 0])[0]
            nm = f"pool{cnt['c']}"; cnt["c"] += 1
            src = ir_of[n.inputs[0]]
            closed.add(src)
            c_src = next(x for x in net.nodes if x.name == src)
            net.add(MaxPool(nm, src, getattr(c_src, "cout", getattr(c_src, "c", 0)), k=k, stride=s, pad=p))
            ir_of[n.outputs[0]] = nm
        elif op in ("AveragePool", "GlobalAveragePool"):
            k = n.attrs.get("kernel_shape", [0])[0]; s = n.attrs.get("strides", [1, 1])[0]
            assert op == "AveragePool" and k == 2 and s == 2, f"{n.name}: only the 2x2/2 average pool of avg_down is supported"
            pending_pool[n.outputs[0]] = n.inputs[0]
            closed.add(ir_of[n.inputs[0]])
        elif op in ("Resize", "Upsample"):
            pending_up[n.outputs[0]] = n.inputs[0]
            closed.add(ir_of[n.inputs[0]])
        elif op in ("Flatten", "Reshape") and any(c.op in ("Gemm", "MatMul") for c in consumers.get(n.outputs[0], [])):
            pending_pool.pop(n.inputs[0], None)
            ir_of[n.outputs[0]] = ir_of.get(n.inputs[0], None)
            if n.inputs[0] in pending_bn:
                pending_bn[n.outputs[0]] = pending_bn[n.inputs[0]]
        elif op in ("Gemm", "MatMul"):
            x = n.inputs[0]
            pre = pending_bn.get(x)
            src = ir_of[pre.inputs[0]] if pre is not None else ir_of[x]
            closed.add(src)
            W = np.asarray(inits[n.inputs[1]], dtype=np.float32)
            if op == "MatMul" or not int(n.attrs.get("transB", 0)):
                W = W.T
            src_node = next(z for z in net.nodes if z.name == src)
            from .archs import infer_shapes
            c_, h_, w_ = infer_shapes(net)[src]
            nm = "fc"
            fc = FC(nm, src, c_, h_, w_, W.shape[0], bias=len(n.inputs) > 2, pre_bn=pre is not None, post_bn=False)
            P[nm + ".weight"] = np.ascontiguousarray(W)
            if len(n.inputs) > 2:
                P[nm + ".bias"] = np.asarray(inits[n.inputs[2]], dtype=np.float32)
            if pre is not None:
                for k2, v2 in _bn_params(inits, pre).items():
                    P[f"{nm}.pre_bn.{k2}"] = v2
            nxt = consumers.get(n.outputs[0], [])
            if len(nxt) == 1 and nxt[0].op == "BatchNormalization":
                fc.post_bn = True
                for k2, v2 in _bn_params(inits, nxt[0]).items():
                    P[f"{nm}.post_bn.{k2}"] = v2
                ir_of[nxt[0].outputs[0]] = nm
            net.add(fc)
            ir_of[n.outputs[0]] = nm
        elif op in ("Sigmoid", "Mul", "Transpose", "Reshape", "Shape", "Gather", "Unsqueeze", "Concat", "Constant",
                    "Identity", "Cast", "Slice", "Squeeze", "Flatten"):
            # detector output plumbing: pass the IR tensor through, remember scalar multipliers and sigmoids
            if op == "Constant":
                inits[n.outputs[0]] = np.asarray(n.attrs.get("value"))
                continue
            x = n.inputs[0]
            if x not in ir_of:
                if op == "Mul" and len(n.inputs) > 1 and n.inputs[1] in ir_of:
                    x = n.inputs[1]
                else:
                    continue
            src = ir_of[x]
            if op == "Mul":
                other = [i for i in n.inputs if i != x][0]
                if other in inits and np.asarray(inits[other]).size == 1:
                    P[src + ".__mul__"] = np.asarray(inits[other], dtype=np.float32).reshape(1)
            if op == "Sigmoid":
                P[src + ".__sigmoid__"] = np.ones(1, np.float32)
            ir_of[n.outputs[0]] = src
        else:
            raise ValueError(f"unsupported ONNX op {op} ({n.name})")

    # ---- outputs -------------------------------------------------------------------------------
    outs = [ir_of[o] for o in g_out]
    if len(outs) == 1:
        net.outputs = outs
        if in_scale is None:
            net.in_scale = 1.0 / 127.5
        return _rename(net, P), _renamed_params(net, P)
    # detector: 9 outputs = 3 strides x (score, bbox, kps); group the output convs by their source tensor
    assert len(outs) == 9, f"expected 1 or 9 graph outputs, got {len(outs)}"
    by_src: Dict[str, List[Conv]] = {}
    for o in outs:
        c = conv_of[o]
        by_src.setdefault(c.src, []).append(c)
    from .archs import infer_shapes
    shapes = infer_shapes(net)
    heads = []
    for src, cs in by_src.items():
        cs = sorted(cs, key=lambda c: c.cout)
        assert [c.cout % 15 == 0 or True for c in cs] and len(cs) == 3, "each stride needs cls/bbox/kps convs"
        A = cs[0].cout
        assert cs[1].cout == 4 * A and cs[2].cout == 10 * A
        stride = net.in_hw[0] // shapes[src][1]
        heads.append((stride, src, cs, A))
    heads.sort(key=lambda h: h[0])
    for stride, src, cs, A in heads:
        nm = f"head.s{stride}.out"
        pos = min(net.nodes.index(c) for c in cs)
        for c, part in zip(cs, ("cls", "bbox", "kps")):
            assert c.k == cs[0].k and not c.post_bn and c.act == "none" and c.res is None
            P[f"{nm}.{part}.weight"] = P.pop(c.name + ".weight")
            P[f"{nm}.{part}.bias"] = P.pop(c.name + ".bias") if c.bias else np.zeros(c.cout, np.float32)
            if part == "bbox":
                P[f"{nm}.bbox.scale"] = P.pop(c.name + ".__mul__", np.ones(1, np.float32))
            P.pop(c.name + ".__mul__", None)
            P.pop(c.name + ".__sigmoid__", None)
            net.nodes.remove(c)
        net.nodes.insert(pos, DetHead(nm, src, cs[0].cin, stride, num_anchors=A, k=cs[0].k))
    net.outputs = [f"head.s{h[0]}.out" for h in heads]
    return _rename(net, P), _renamed_params(net, P)


def _rename(net, P):
    return net


def _renamed_params(net, P):
    return {k: v for k, v in P.items() if "__" not in k}


_KNOWN = {"det_10g": ("scrfd_10g", 1 / 128.0), "det_2.5g": ("scrfd_2.5g", 1 / 128.0), "det_500m": ("scrfd_500m", 1 / 128.0),
          "w600k_r50": ("arcface_r50", 1 / 127.5), "w600k_mbf": ("arcface_mbf", 1 / 127.5)}


def load_onnx_model(path: str, in_hw=None):
    """(Net, params) for a .onnx file; the normalisation constants come from the reference's call sites
    (scrfd.py:44-45,76-82: mean 127.5, 1/128; arcface.py:13-14,44-50: mean 127.5, 1/127.5)."""
    import os
    base = os.path.splitext(os.path.basename(path))[0]
    name, scale = _KNOWN.get(base, (base, None))
    data = open(path, "rb").read()
    return onnx_to_ir(data, name, in_hw=in_hw, in_scale=scale)

"""Lowering: archs.Net + raw fp32 parameters  ->  the layer table libfaceid executes.

This is the work onnxruntime's graph optimiser does invisibly inside InferenceSession (reference
models/scrfd.py:59-62, models/arcface.py:18-21), written down:

  * BatchNorm after a conv is folded into the conv's weights and bias;
  * BatchNorm BEFORE a zero-padded conv (IResNet's bn1) is folded exactly: scale into the weights,
    shift into a 9-entry border-class bias table (a border pixel's missing taps contribute no shift);
  * AvgPool(2)+Conv1x1 ("avg_down" shortcut) becomes one 2x2/stride-2 conv with weights/4;
  * nearest-2x upsample + add (PAFPN top-down) and the residual add become epilogue flags;
  * a residual BasicBlock on 64 (padded) channels -- conv3x3+ReLU, conv3x3, + block input, activation -- becomes ONE op whose
    intermediate map never leaves the CU (csrc/conv_bb.hip; FID_NO_BB_FUSE=1 keeps the two convs);
  * a depthwise 3x3 conv whose only consumer is a pointwise 1x1 conv (MobileFaceNet bottlenecks, SCRFD-500M) becomes ONE op with the
    depthwise result in LDS (csrc/dwpw.hip) when FID_DWPW_FUSE=1 asks for it (measured slower than the two launches: off by default);
  * the three SCRFD output convs of a level become one conv with 2+8+20 output channels, sigmoid on
    the first two, bbox scale folded in, fp32 output;
  * blobFromImage's (x-127.5)*scale and BGR->RGB swap are folded into the first conv's weights;
  * a PAFPN level's 1x1 lateral conv (+ the upsampled coarser lateral) and the 3x3 conv on it become ONE op (csrc/lat_fpn.hip): the lateral is
    stored only when a finer level adds it (FID_NO_LATFPN_FUSE=1 keeps the two convs apart);
  * IResNet's first conv (3 -> 64, stride 1) and the 3x3 conv on 64 channels that consumes it become ONE op (csrc/stem_block.hip): the first
    conv's map only exists in LDS; the block's 1x1 / stride-2 shortcut reads a compact copy of it at the even pixels (FID_NO_STEMBLOCK_FUSE=1
    keeps the two convs apart);
  * FC consumes the NHWC activation directly (weight columns permuted from CHW order).

Weights are packed fp16 [Cout_p][tap][Cin_p] (channels padded to multiples of 32 with zeros), all
epilogue tables fp32.  Activation buffers are assigned to slots by liveness.
Word layout of the op / tensor records: csrc/net.h.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

from .archs import BN_EPS, Net, infer_shapes

OP_WORDS, TENSOR_WORDS = 32, 8
OP_STEM, OP_CONV, OP_MAXPOOL, OP_DWCONV, OP_STEMFUSED, OP_BBLOCK, OP_DWPW, OP_MBBLOCK, OP_STEMBLOCK, OP_LATFPN = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10
ACT = {"none": 0, "relu": 1, "prelu": 2}
CF_RES_UP2, CF_BORDER, CF_OUT_F32 = 1, 2, 4
CPAD = 32


def _rup(x, m):
    return (x + m - 1) // m * m


def _bn_affine(P, prefix):
    a = P[prefix + ".gamma"].astype(np.float64) / np.sqrt(P[prefix + ".var"].astype(np.float64) + BN_EPS)
    b = P[prefix + ".beta"].astype(np.float64) - P[prefix + ".mean"].astype(np.float64) * a
    return a, b


class _Blob:
    def __init__(self):
        self.parts: List[bytes] = []
        self.size = 0

    def add(self, arr: np.ndarray) -> Tuple[int, int]:
        raw = np.ascontiguousarray(arr).tobytes()
        off = self.size
        pad = (-len(raw)) % 256
        self.parts.append(raw + b"\0" * pad)
        self.size += len(raw) + pad
        return off, len(raw)

    def bytes(self) -> bytes:
        return b"".join(self.parts)


class Lowered:
    """Result of lower(): numpy tables + name maps."""

    def __init__(self):
        self.ops: np.ndarray = None          # int32 [n_ops, 32]
        self.tensors: np.ndarray = None      # int32 [n_tensors, 8]
        self.blob: bytes = b""
        self.tensor_id: Dict[str, int] = {}
        self.op_names: List[str] = []
        self.op_nodes: List[List[str]] = []
        self.outputs: List[str] = []
        self.in_hw = (0, 0)
        self.macs = 0
        self.heads: Dict[str, dict] = {}     # DetHead name -> channel layout of the fused tensor
        self.fused_groups: Dict[str, list] = {}   # fused op name -> graph nodes it covers


def _dual_ok(net, i, tensors, tid, shp):
    """nodes i, i+1 = an "avg_down" shortcut conv and the block's 3x3 / stride-2 conv1 on the same input, in the shapes csrc/conv_s2.hip's DUAL
    variant takes (64 stored input channels, 96 padded couts each): lowered as ONE op with two outputs"""
    import os
    n = net.nodes[i]
    nxt = net.nodes[i + 1] if i + 1 < len(net.nodes) else None
    return (n.kind == "conv" and n.groups == 1 and n.pre_avgpool and not os.environ.get("FID_NO_DOWN_FUSE")
            and nxt is not None and nxt.kind == "conv" and nxt.groups == 1 and nxt.src == n.src and nxt.k == 3 and nxt.stride == 2
            and nxt.pad == 1 and nxt.res is None and not nxt.pre_bn and not n.pre_bn and not nxt.pre_avgpool
            and nxt.act in ("none", "relu") and n.act == "none" and n.res is None and not n.res_up2 and not nxt.res_up2
            and tensors[tid[n.src]][1] == 64 and _rup(n.cout, CPAD) == 96 and _rup(nxt.cout, CPAD) == 96
            and shp[n.name][1:] == shp[nxt.name][1:])


def _shortcut_target(net, i, tensors, tid, shp):
    """node i = a residual block's shortcut conv whose only consumer is the residual add of a later 3x3 conv of the same output shape that the
    implicit-GEMM kernel can extend by K-steps on the block input: that conv node, or None.  Two forms:
      * IResNet: 1x1 / stride s + BN feeding the block's stride-s conv2 (one extra tap at (s oy, s ox));
      * ResNetV1e "avg_down" (SCRFD): 2x2 average pool + 1x1 + BN = a 2x2 / stride-2 conv feeding the block's stride-1 conv2 (four extra taps at
        (2 oy + {0,1}, 2 ox + {0,1})) -- unless csrc/conv_s2.hip's DUAL launch already computes it beside conv1 (_dual_ok)."""
    n = net.nodes[i]
    if not (n.k == 1 and n.pad == 0 and n.act == "none" and n.res is None and not n.pre_bn and not n.res_up2 and n.src != "input"
            and n.name not in net.outputs and n.groups == 1):
        return None
    avg = bool(n.pre_avgpool)
    if (avg and n.stride != 1) or (not avg and n.stride not in (1, 2)) or (avg and _dual_ok(net, i, tensors, tid, shp)):
        return None
    s_eff, taps = (2, 2) if avg else (n.stride, 1)
    users = [x for x in net.nodes if getattr(x, "src", None) == n.name or getattr(x, "res", None) == n.name]
    if len(users) != 1:
        return None
    m = users[0]
    if not (m.kind == "conv" and m.res == n.name and m.src != n.name and m.groups == 1 and m.k == 3 and m.pad == 1 and not m.pre_bn
            and m.stride == (1 if avg else 2) and not m.pre_avgpool and not m.res_up2 and net.nodes.index(m) > i and shp[m.name] == shp[n.name]):
        return None
    x_t = tensors[tid[n.src]]
    if m.src not in tid:                                  # (conv2's input is lowered after the shortcut: its padded width is its own cout rounded up)
        src_node = next(x for x in net.nodes if x.name == m.src)
        cin_p = _rup(src_node.cout, CPAD)
    else:
        cin_p = tensors[tid[m.src]][1]
    # every sampled pixel (oy * s + dy, ox * s + dx) must exist, the channel counts must suit a 32- or 64-wide K-step of the LDS-DMA implicit GEMM
    _, ho, wo = shp[m.name]
    if x_t[4] != 0 or x_t[1] % 32 or cin_p % 32 or (ho - 1) * s_eff + taps - 1 >= x_t[2] or (wo - 1) * s_eff + taps - 1 >= x_t[3]:
        return None
    if not avg and n.stride != m.stride and not getattr(n, "_even_src", False):   # (_even_src: the shortcut reads the fused stem block's even-pixel copy of x with stride 1)
        return None
    return m


def _mbf_block(net, i, tensors, tid, shp):
    """nodes i, i+1, i+2 = pointwise 1x1 -> depthwise 3x3 -> pointwise 1x1 [+ the first one's input] in the shapes csrc/mbf_block.hip takes
    (mbf_block_applicable): (depthwise node, second pointwise node) or None"""
    if i + 2 >= len(net.nodes):
        return None
    n, d, m = net.nodes[i], net.nodes[i + 1], net.nodes[i + 2]
    if not (d.kind == "conv" and m.kind == "conv" and n.k == 1 and n.pad == 0 and n.stride == 1 and n.src != "input" and n.res is None
            and not n.pre_bn and not n.pre_avgpool and not n.res_up2
            and d.groups == d.cin == d.cout == n.cout and d.groups > 1 and d.k == 3 and d.pad == 1 and d.stride in (1, 2) and d.src == n.name
            and d.res is None and not d.pre_bn and not d.pre_avgpool
            and m.groups == 1 and m.k == 1 and m.pad == 0 and m.stride == 1 and m.src == d.name and m.res in (None, n.src)
            and not m.pre_bn and not m.pre_avgpool and not m.res_up2):
        return None
    if n.name in net.outputs or d.name in net.outputs:
        return None
    for x in net.nodes:                                   # the two expanded maps feed nothing else
        if x is not d and (getattr(x, "src", None) == n.name or getattr(x, "res", None) == n.name):
            return None
        if x is not m and (getattr(x, "src", None) == d.name or getattr(x, "res", None) == d.name):
            return None
    src_t = tensors[tid[n.src]]
    cin_p, gp, cout_p, H, W = src_t[1], _rup(n.cout, CPAD), _rup(m.cout, CPAD), src_t[2], src_t[3]
    if src_t[4] != 0 or cin_p % 32 or cin_p > 256 or gp % 32 or cout_p % 16 or cout_p > 256:
        return None
    if m.res is not None and (d.stride != 1 or cin_p != cout_p):
        return None
    if gp > 512:
        return None
    to = 7 if d.stride == 1 else 4                        # LDS: the 9 x 9 region + both expanded maps (csrc/mbf_block.hip mbf_lds_bytes; the kernel
    rup = lambda v: (v + 1023) // 1024 * 1024            #  takes 7 x 7 tiles at stride 2 as well when their 15 x 15 region fits)
    if rup(81 * (cin_p * 2 + 16)) + rup(81 * (gp * 2 + 16)) + ((to * to + 15) // 16 * 16) * (gp * 2 + 16) > 160 * 1024:
        return None
    return d, m


def _latfpn(net, i, tensors, tid):
    """node i = a 1x1 lateral conv (bias / BN, no activation, optionally + the nearest-2x upsampled coarser lateral) on 64 / 96 stored channels whose
    consumers are ONE 3x3 / stride-1 conv without activation or residual (64 stored couts each) and otherwise only finer laterals that add it
    (res_up2): (index of the 3x3 conv, that node, lateral stored?) or None (csrc/lat_fpn.hip)"""
    import os
    n = net.nodes[i]
    if os.environ.get("FID_NO_LATFPN_FUSE") or not (n.kind == "conv" and n.k == 1 and n.pad == 0 and n.stride == 1 and n.groups == 1 and n.act == "none"
                                                    and not n.pre_bn and not n.pre_avgpool and n.src != "input" and n.name not in net.outputs
                                                    and (n.res is None or n.res_up2) and _rup(n.cout, CPAD) == 64):
        return None
    src_t = tensors[tid[n.src]]
    if src_t[4] != 0 or src_t[1] not in (64, 96) or src_t[2] < 3 or src_t[3] < 3:
        return None
    if n.res is not None:
        r_t = tensors[tid[n.res]]
        if r_t[4] != 0 or r_t[1] != 64 or src_t[2] != 2 * r_t[2] or src_t[3] != 2 * r_t[3]:      # exact 2x levels only (see the CF_RES_UP2 note below)
            return None
    convs, adders = [], 0
    for j, x in enumerate(net.nodes):
        if getattr(x, "src", None) == n.name:
            convs.append((j, x))
        if getattr(x, "res", None) == n.name:
            if not (x.kind == "conv" and x.res_up2 and x.src != n.name):
                return None
            adders += 1
    if len(convs) != 1:
        return None
    j, m = convs[0]
    if not (j > i and m.kind == "conv" and m.k == 3 and m.stride == 1 and m.pad == 1 and m.groups == 1 and m.act == "none" and m.res is None and not m.pre_bn
            and not m.pre_avgpool and m.cin == n.cout and _rup(m.cout, CPAD) == 64):
        return None
    return j, m, adders > 0


def lower(net: Net, P: Dict[str, np.ndarray]) -> Lowered:
    shp = infer_shapes(net)
    blob = _Blob()
    ops: List[List[int]] = []
    op_names: List[str] = []
    op_nodes: List[List[str]] = []        # graph nodes each op covers (several for fused ops)
    tensors: List[List[int]] = []
    tid: Dict[str, int] = {}
    out = Lowered()

    def new_tensor(name, C, H, W, dtype=0, Cp=None):
        Cp = _rup(C, CPAD) if Cp is None else Cp
        tid[name] = len(tensors)
        tensors.append([C, Cp, H, W, dtype, -1, 0, 0])
        return tid[name]

    def emit(name, **kw):
        rec = [0] * OP_WORDS
        rec[3] = -1
        rec[13] = rec[15] = rec[16] = -1
        rec[18] = 1
        idx = dict(type=0, src=1, dst=2, res=3, kh=4, kw=5, stride=6, pad=7, cin=8, cout=9, act=10, flags=11,
                   nsig=12, woff=13, wbytes=14, boff=15, soff=16, wrows=17, groups=18)
        for k, v in kw.items():
            rec[idx[k]] = int(v)
        ops.append(rec)
        op_names.append(name)
        op_nodes.append([name])

    def pack_weights(W4, cin_p, cout_p):
        """[cout, cin, kh, kw] float64 -> fp16 [cout_p][kh*kw][cin_p]"""
        cout, cin, kh, kw = W4.shape
        Wp = np.zeros((cout_p, kh * kw, cin_p), dtype=np.float32)
        Wp[:cout, :, :cin] = W4.transpose(0, 2, 3, 1).reshape(cout, kh * kw, cin)
        h = Wp.astype(np.float16)
        if not np.isfinite(h).all():
            raise ValueError("weights overflow fp16")
        return h

    def repack_kind2(Wp):
        """fp16 [cout_p <= 128][9][cin_p] -> csrc/repack.hip kind 2 (conv3x3_wr's MFMA A fragments in lane order):
        [32-channel chunk][cout fragment 0..7][dx][dy][lane][8 halfs], lane = (channel group of 8) * 16 + (cout & 15)"""
        cout_p, taps, cin_p = Wp.shape
        assert taps == 9 and cout_p <= 128 and cin_p % 32 == 0
        full = np.zeros((128, 9, cin_p), dtype=np.float16)
        full[:cout_p] = Wp
        x = full.reshape(8, 16, 3, 3, cin_p // 32, 4, 8)          # [cf][cout & 15][dy][dx][ck][group][e]
        return np.ascontiguousarray(x.transpose(4, 0, 3, 2, 5, 1, 6))   # [ck][cf][dx][dy][group][cout & 15][e]

    def padded(vec, n, fill=0.0):
        o = np.full(n, fill, dtype=np.float32)
        o[: len(vec)] = vec
        return o

    def folded(n):
        """conv weights/bias with the trailing BatchNorm folded in (float64)"""
        W = P[n.wname + ".weight"].astype(np.float64)
        b = P[n.wname + ".bias"].astype(np.float64) if n.bias else np.zeros(n.cout)
        if n.post_bn:
            a2, b2 = _bn_affine(P, n.wname + ".post_bn")
            W = W * a2[:, None, None, None]
            b = b * a2 + b2
        return W, b

    def fold_conv(n, src_hw):
        """full fold of a conv node: (W [cout, cin, k, k] f64, bias table [ncls, cout] f64, k, stride) with a BatchNorm in FRONT of a zero-padded
        conv folded exactly (scale into the weights, shift into 9 border-class bias rows: a border pixel's missing taps contribute no shift), the
        avg-pool of an "avg_down" shortcut as a 2x2 / stride-2 kernel and the trailing BatchNorm"""
        W = P[n.wname + ".weight"].astype(np.float64)            # [cout, cin, k, k]
        b = P[n.wname + ".bias"].astype(np.float64) if n.bias else np.zeros(n.cout)
        k, stride, pad = n.k, n.stride, n.pad
        if n.pre_avgpool:
            assert k == 1 and pad == 0 and stride == 1
            W = np.repeat(np.repeat(W, 2, axis=2), 2, axis=3) / 4.0
            k, stride = 2, 2
        bias_tab = None
        if n.pre_bn:
            a1, b1 = _bn_affine(P, n.wname + ".pre_bn")
            shift = np.einsum("oikl,i->okl", W, b1)                # per-tap contribution of the BN shift
            W = W * a1[None, :, None, None]
            if pad == 0:
                b = b + shift.sum(axis=(1, 2))
            else:
                assert k == 3 and pad == 1 and stride == 1 and src_hw[0] >= 2 and src_hw[1] >= 2, n.name
                bias_tab = np.zeros((9, n.cout))
                for yc in range(3):
                    for xc in range(3):
                        m = np.ones((3, 3))
                        if yc == 0: m[0, :] = 0
                        if yc == 2: m[2, :] = 0
                        if xc == 0: m[:, 0] = 0
                        if xc == 2: m[:, 2] = 0
                        bias_tab[yc * 3 + xc] = b + (shift * m[None]).sum(axis=(1, 2))
        if bias_tab is None:
            bias_tab = b[None, :]
        if n.post_bn:
            a2, b2 = _bn_affine(P, n.wname + ".post_bn")
            W = W * a2[:, None, None, None]
            bias_tab = bias_tab * a2[None, :] + b2[None, :]
        return W, bias_tab, k, stride

    # ---- SCRFD deep stem (conv/s2 - conv - conv - maxpool, all ReLU): one fused kernel (csrc/stem_fused.hip) ----
    import os
    fused_upto = 0
    nd = net.nodes
    if (len(nd) >= 4 and not os.environ.get("FID_NO_STEM_FUSE") and nd[0].kind == "conv" and nd[0].src == "input"
            and all(x.kind == "conv" and x.k == 3 and x.pad == 1 and x.groups == 1 and x.act == "relu" and not x.pre_bn
                    and x.res is None and not x.pre_avgpool for x in nd[0:3])
            and nd[0].stride == 2 and nd[1].stride == 1 and nd[2].stride == 1 and nd[1].src == nd[0].name
            and nd[2].src == nd[1].name and nd[3].kind == "maxpool" and nd[3].src == nd[2].name
            and (nd[3].k, nd[3].stride, nd[3].pad) == (3, 2, 1) and nd[0].cout <= 32 and nd[1].cout <= 32
            and nd[2].cout <= 64 and net.in_hw[0] % 4 == 0 and net.in_hw[1] % 4 == 0
            and not any(getattr(x, "src", None) in (nd[0].name, nd[1].name, nd[2].name) or getattr(x, "res", None) in
                        (nd[0].name, nd[1].name, nd[2].name) for x in nd[4:])
            and not any(o in (nd[0].name, nd[1].name, nd[2].name) for o in net.outputs)):
        c0, c1, c2, pool = nd[0], nd[1], nd[2], nd[3]
        W0, b0 = folded(c0)
        W1, b1 = folded(c1)
        W2, b2 = folded(c2)
        w0 = np.zeros((32, 32), dtype=np.float32)                      # [co][k = (dy*3+dx)*3 + c_bgr]
        w0[: c0.cout, :27] = (W0[:, ::-1] * (net.in_scale / 2.0)).transpose(0, 2, 3, 1).reshape(c0.cout, 27)
        c2p = _rup(c2.cout, CPAD)
        offs = [blob.add(w0.astype(np.float16))[0], blob.add(padded(b0, 32))[0],
                blob.add(pack_weights(W1, 32, 32))[0], blob.add(padded(b1, 32))[0],
                blob.add(pack_weights(W2, 32, c2p))[0], blob.add(padded(b2, c2p))[0]]
        _, hp, wp = shp[pool.name]
        dst = new_tensor(pool.name, c2.cout, hp, wp)
        rec = [0] * OP_WORDS
        rec[0], rec[1], rec[2], rec[3] = OP_STEMFUSED, -1, dst, -1
        rec[4] = rec[5] = 3
        rec[13] = rec[15] = rec[16] = -1
        rec[18] = 1
        rec[20:26] = offs
        macs = sum(shp[x.name][1] * shp[x.name][2] * x.cout * x.cin * 9 for x in (c0, c1, c2))
        rec[26], rec[27] = macs & 0xFFFFFFFF, macs >> 32
        if rec[26] >= 2 ** 31:
            rec[26] -= 2 ** 32
        ops.append(rec)
        op_names.append("stem.fused")
        op_nodes.append([c0.name, c1.name, c2.name, pool.name])
        out.fused_groups = {"stem.fused": [c0.name, c1.name, c2.name, pool.name]}
        fused_upto = 4

    skip = set()
    # ---- IResNet's stem + the first 3x3 conv on its result: one fused op (csrc/stem_block.hip) ----
    if fused_upto == 0 and not os.environ.get("FID_NO_STEMBLOCK_FUSE") and len(nd) >= 2:
        st = nd[0]
        users = [(i, x) for i, x in enumerate(nd) if getattr(x, "src", None) == st.name or getattr(x, "res", None) == st.name]
        c1s = [(i, x) for i, x in users if x.kind == "conv" and x.src == st.name and x.k == 3 and x.stride == 1 and x.pad == 1 and x.groups == 1
               and x.cin == 64 and x.cout == 64 and x.res is None and x.act in ("relu", "prelu") and not x.pre_avgpool and not x.res_up2]
        others = [(i, x) for i, x in users if not c1s or x is not c1s[0][1]]
        sc_ok = all(x.kind == "conv" and x.src == st.name and x.res != st.name and x.k == 1 and x.stride == 2 and x.pad == 0 and x.groups == 1
                    and not x.pre_bn and not x.pre_avgpool for _, x in others)
        H0, W0 = net.in_hw
        if (st.kind == "conv" and st.src == "input" and st.k == 3 and st.stride == 1 and st.pad == 1 and st.groups == 1 and st.cout == 64
                and st.act in ("relu", "prelu") and st.res is None and not st.pre_bn and len(c1s) == 1 and sc_ok and len(others) <= 1
                and st.name not in net.outputs and W0 % 4 == 0 and H0 >= 3 and abs(net.in_mean - 127.5) < 1e-12):
            import copy
            import dataclasses
            ci, c1 = c1s[0]
            W = P[st.wname + ".weight"].astype(np.float64)               # [64, 3(RGB), 3, 3]
            b = P[st.wname + ".bias"].astype(np.float64) if st.bias else np.zeros(64)
            if st.post_bn:
                a2, b2 = _bn_affine(P, st.wname + ".post_bn")
                W = W * a2[:, None, None, None]
                b = b * a2 + b2
            Wd = np.ascontiguousarray((W[:, ::-1] * (net.in_scale / 2.0)).transpose(0, 2, 3, 1), dtype=np.float32)   # [co][dy][dx][c_bgr]: the kernel's input is 2 p - 255
            w0off, w0bytes = blob.add(Wd)
            b0off = blob.add(padded(b, 64))[0]
            s0off = blob.add(padded(P[st.wname + ".prelu"], 64))[0] if st.act == "prelu" else -1
            W1, bt1, _, _ = fold_conv(c1, (H0, W0))
            # stem_block.hip is built with -fno-honor-nans (its max(v, 0) then needs no canonicalising second v_max): a folded weight or bias that
            # is inf / NaN, or overflows fp16, would be undefined behaviour there instead of propagating -- refused here
            for nm_, arr_ in (("stem weights", Wd), ("stem bias", b), ("conv1 weights", W1), ("conv1 bias", bt1)):
                if not np.isfinite(np.asarray(arr_, dtype=np.float64)).all() or np.abs(np.asarray(arr_, dtype=np.float64)).max() > 65504.0:
                    raise ValueError(f"{st.name} / {c1.name}: folded {nm_} are not finite fp16 values; the fused stem block cannot take them")
            b1t = np.zeros((bt1.shape[0], 64), dtype=np.float32)
            b1t[:, :64] = bt1
            w1off = blob.add(repack_kind2(pack_weights(W1, 64, 64)))[0]
            b1off = blob.add(b1t)[0]
            s1off = blob.add(padded(P[c1.wname + ".prelu"], 64))[0] if c1.act == "prelu" else -1
            dst = new_tensor(c1.name, 64, H0, W0)
            dst2 = 0
            if others:                                                  # the block's stride-2 shortcut reads x at the even pixels only: a compact second output
                even = st.name + ".even"
                dst2 = new_tensor(even, 64, (H0 + 1) // 2, (W0 + 1) // 2) + 1
                shp[even] = (64, (H0 + 1) // 2, (W0 + 1) // 2)
                oi_, sc_node = others[0]
                sc2 = dataclasses.replace(sc_node, src=even, stride=1)
                sc2._even_src = True
                net = copy.copy(net)
                net.nodes = list(nd)
                net.nodes[oi_] = sc2
                nd = net.nodes
            rec = [0] * OP_WORDS
            rec[0], rec[1], rec[2], rec[3] = OP_STEMBLOCK, -1, dst, -1
            rec[4] = rec[5] = 3
            rec[6], rec[7], rec[8], rec[9], rec[10] = 1, 1, 3, 64, ACT[c1.act]
            rec[11] = CF_BORDER if bt1.shape[0] == 9 else 0
            rec[13], rec[14], rec[15], rec[16], rec[17], rec[18] = w0off, w0bytes, b0off, s0off, 64, 1
            rec[20], rec[21], rec[22], rec[23], rec[24] = w1off, b1off, s1off, ACT[st.act], dst2
            macs = H0 * W0 * 64 * 27 + H0 * W0 * 64 * 64 * 9
            rec[26], rec[27] = macs & 0xFFFFFFFF, macs >> 32
            if rec[26] >= 2 ** 31:
                rec[26] -= 2 ** 32
            ops.append(rec)
            op_names.append(c1.name)
            op_nodes.append([st.name, c1.name])
            out.fused_groups[c1.name] = [st.name, c1.name]
            skip.update({0, ci})
    pending_sc = {}                                   # conv2 name -> its block's shortcut conv node, fused as extra K-steps (below)
    for ni_, n in enumerate(net.nodes):
        if ni_ < fused_upto or ni_ in skip:
            continue
        nxt = net.nodes[ni_ + 1] if ni_ + 1 < len(net.nodes) else None
        if (n.kind == "conv" and n.groups == 1 and n.src != "input" and not os.environ.get("FID_NO_SC_FUSE")
                and _shortcut_target(net, ni_, tensors, tid, shp) is not None):
            # IResNet's block shortcut (1x1 / stride-2 conv + BN on the block input x) feeds nothing but the residual add of the block's stride-2
            # conv2: it CAN run as extra K-steps of that conv (one more "tap" that reads x at (2 oy, 2 ox): csrc/conv.hip generation 2,
            # ConvArgs::in2).  Both forms are lowered -- this op as it is, and conv2 with a second weight image [9 Cin | Cx] + summed bias --
            # and the autotuner decides per batch size (the fused form is a generation-12 pick of conv2; this op is then skipped): at 64
            # faces every fused pair wins 5-12 us, at 500 the weights-in-registers kernels the fused form cannot use win instead.
            pending_sc[_shortcut_target(net, ni_, tensors, tid, shp).name] = (n, len(ops))     # (this op is emitted next: its index)
        if n.kind == "conv" and n.src == "input":
            assert n.k == 3 and n.pad == 1 and n.groups == 1 and not n.pre_bn and n.res is None
            cout, (_, ho, wo) = n.cout, shp[n.name]
            cp = _rup(cout, CPAD)
            assert cp in (32, 64, 128), cp
            W = P[n.wname + ".weight"].astype(np.float64)           # [cout, 3(RGB), 3, 3]
            b = P[n.wname + ".bias"].astype(np.float64) if n.bias else np.zeros(cout)
            if n.post_bn:
                a2, b2 = _bn_affine(P, n.wname + ".post_bn")
                W = W * a2[:, None, None, None]
                b = b * a2 + b2
            # kernel input is (2*pixel - 255) in BGR order: fold scale/2 and the channel swap
            Wd = np.zeros((cp, 3, 3, 3), dtype=np.float32)            # [co][dy][dx][c_bgr]
            Wd[:cout] = (W[:, ::-1] * (net.in_scale / 2.0)).transpose(0, 2, 3, 1)
            assert abs(net.in_mean - 127.5) < 1e-12
            woff, wbytes = blob.add(Wd)
            boff, _ = blob.add(padded(b, cp))
            soff = blob.add(padded(P[n.wname + ".prelu"], cp))[0] if n.act == "prelu" else -1
            dst = new_tensor(n.name, cout, ho, wo)
            emit(n.name, type=OP_STEM, src=-1, dst=dst, kh=3, kw=3, stride=n.stride, pad=1, cin=3, cout=cout,
                 act=ACT[n.act], woff=woff, wbytes=wbytes, boff=boff, soff=soff, wrows=cp)
        elif _dual_ok(net, ni_, tensors, tid, shp):
            # The block's shortcut (2x2 average pool + 1x1 conv + BN = a 2x2 / stride-2 conv with W/4) reads the tensor the block's
            # first conv (3x3 / stride 2 / pad 1) reads, and its window is taps (1..2, 1..2) of that conv's window: ONE launch of the
            # stride-2 kernel with twice the couts and two outputs fetches the (large) input once instead of twice (csrc/conv_s2.hip,
            # twelve waves: fragments 0..5 = the conv, 6..11 = the shortcut).  Op record: the conv's, with 192 weight / bias rows;
            # word 20 = second output's tensor id + 1, 21 = its activation, 22 = padded couts of the first output, 26 / 27 = the MACs
            # of the shortcut (cost accounting).
            m = nxt
            src_t = tensors[tid[n.src]]
            cin_p = src_t[1]
            _, ho, wo = shp[m.name]

            def folded_w(c, W):
                b = P[c.wname + ".bias"].astype(np.float64) if c.bias else np.zeros(c.cout)
                if c.post_bn:
                    a2, b2 = _bn_affine(P, c.wname + ".post_bn")
                    W = W * a2[:, None, None, None]
                    b = b * a2 + b2
                return W, b
            W1, b1 = folded_w(m, P[m.wname + ".weight"].astype(np.float64))                 # [cout, cin, 3, 3]
            Wd = P[n.wname + ".weight"].astype(np.float64)                                # [cout, cin, 1, 1]
            Wd = np.repeat(np.repeat(Wd, 2, axis=2), 2, axis=3) / 4.0                     # avgpool2 folded: 2x2 / stride 2
            Wd, b2 = folded_w(n, Wd)
            W2 = np.zeros((n.cout, n.cin, 3, 3))
            W2[:, :, 1:3, 1:3] = Wd
            Wp = np.concatenate([pack_weights(W1, cin_p, 96), pack_weights(W2, cin_p, 96)], axis=0)
            woff, wbytes = blob.add(Wp)
            bt = np.zeros((1, 192), dtype=np.float32)
            bt[0, :m.cout] = b1
            bt[0, 96:96 + n.cout] = b2
            boff, _ = blob.add(bt)
            dst2 = new_tensor(n.name, n.cout, ho, wo)
            dst = new_tensor(m.name, m.cout, ho, wo)
            emit(m.name, type=OP_CONV, src=tid[n.src], dst=dst, res=-1, kh=3, kw=3, stride=2, pad=1, cin=m.cin, cout=m.cout,
                 act=ACT[m.act], flags=0, woff=woff, wbytes=wbytes, boff=boff, soff=-1, wrows=192)
            ops[-1][20], ops[-1][21], ops[-1][22] = dst2 + 1, ACT[n.act], 96
            macs2 = ho * wo * n.cout * n.cin * 4
            assert macs2 < 2 ** 31
            ops[-1][26], ops[-1][27] = macs2, 0
            op_nodes[-1] = [n.name, m.name]
            out.fused_groups[m.name] = [n.name, m.name]
            skip.add(ni_ + 1)
        elif (n.kind == "conv" and n.groups == 1 and not os.environ.get("FID_NO_BB_FUSE") and nxt is not None and nxt.kind == "conv"
              and nxt.groups == 1 and n.k == 3 and nxt.k == 3 and n.stride == 1 and nxt.stride == 1 and n.pad == 1 and nxt.pad == 1
              and n.act in ("relu", "prelu") and nxt.act in ("relu", "none") and n.res is None and nxt.res == n.src and nxt.src == n.name
              and not n.res_up2 and not nxt.res_up2 and not nxt.pre_bn and not n.pre_avgpool and not nxt.pre_avgpool
              and n.src != "input" and tensors[tid[n.src]][1] in (32, 64) and tensors[tid[n.src]][4] == 0 and _rup(n.cout, CPAD) == tensors[tid[n.src]][1]
              and _rup(nxt.cout, CPAD) == tensors[tid[n.src]][1] and (tensors[tid[n.src]][1] == 64 or (n.act == "relu" and not n.pre_bn))
              and n.name not in net.outputs and tensors[tid[n.src]][2] >= 3 and tensors[tid[n.src]][3] >= 3
              and not any(getattr(x, "src", None) == n.name or getattr(x, "res", None) == n.name for x in net.nodes if x is not nxt)):
            # A residual BasicBlock on 64 stored channels (SCRFD-10G layer1): conv1 + ReLU -> conv2 -> + block input -> activation as ONE
            # launch (csrc/conv_bb.hip): conv1's output only feeds conv2, so it is never materialised.  Both filter banks go into the blob
            # in the register-fragment order the kernel loads (repack.hip kind 2).
            # IResNet's form of the block (BN - conv - BN - PReLU - conv - BN, + input: arcface_r50 layer1.1 / 1.2) is covered too: the leading
            # BatchNorm folds into conv1 with 9 border-class bias rows (flag CF_BORDER), PReLU slopes travel in word 24.
            m = nxt
            src_t = tensors[tid[n.src]]
            W1, bt1, _, _ = fold_conv(n, (src_t[2], src_t[3]))
            W2, bt2, _, _ = fold_conv(m, (src_t[2], src_t[3]))
            assert bt2.shape[0] == 1
            cpb = src_t[1]                                            # 64 stored channels, or 32 (conv_bb32: SCRFD-2.5G layer1)
            b1t = np.zeros((bt1.shape[0], cpb), dtype=np.float32)
            b1t[:, :n.cout] = bt1
            offs = [blob.add(repack_kind2(pack_weights(W1, cpb, cpb)))[0], blob.add(b1t)[0],
                    blob.add(repack_kind2(pack_weights(W2, cpb, cpb)))[0], blob.add(padded(bt2[0], cpb))[0]]
            s1_off = blob.add(padded(P[n.wname + ".prelu"], cpb))[0] if n.act == "prelu" else -1
            _, ho, wo = shp[m.name]
            dst = new_tensor(m.name, m.cout, ho, wo)
            rec = [0] * OP_WORDS
            rec[0], rec[1], rec[2], rec[3] = OP_BBLOCK, tid[n.src], dst, -1
            rec[4] = rec[5] = 3
            rec[6], rec[7] = 1, 1
            rec[8], rec[9], rec[10] = n.cin, m.cout, ACT[m.act]
            rec[13] = rec[15] = rec[16] = -1
            rec[18] = 1
            rec[20:24] = offs
            rec[24], rec[25] = s1_off, ACT[n.act]
            rec[11] = CF_BORDER if bt1.shape[0] == 9 else 0
            macs = ho * wo * 9 * (n.cout * n.cin + m.cout * m.cin)
            rec[26], rec[27] = macs & 0xFFFFFFFF, macs >> 32
            if rec[26] >= 2 ** 31:
                rec[26] -= 2 ** 32
            ops.append(rec)
            op_names.append(m.name)
            op_nodes.append([n.name, m.name])
            out.fused_groups[m.name] = [n.name, m.name]
            skip.add(ni_ + 1)
        elif (n.kind == "conv" and n.groups == 1 and not os.environ.get("FID_NO_MBF_FUSE") and _mbf_block(net, ni_, tensors, tid, shp) is not None):
            # MobileFaceNet's bottleneck: 1x1 (cin -> G, act) -> depthwise 3x3 / stride 1 | 2 (G, act) -> 1x1 (G -> cout) [+ block input] as ONE launch
            # (csrc/mbf_block.hip): both expanded maps stay in LDS.  The record is the second pointwise conv's; words 20-25, 28-30 hold the first
            # pointwise conv's and the depthwise layer's tables (csrc/net.h W_M_*), 26 / 27 the MACs of the three layers.
            d, m = _mbf_block(net, ni_, tensors, tid, shp)
            src_t = tensors[tid[n.src]]
            cin_p, gp, cout_p = src_t[1], _rup(n.cout, CPAD), _rup(m.cout, CPAD)
            W1, b1 = folded(n)
            w1_off = blob.add(pack_weights(W1, cin_p, gp))[0]
            b1_off = blob.add(padded(b1, gp))[0]
            s1_off = blob.add(padded(P[n.wname + ".prelu"], gp))[0] if n.act == "prelu" else -1
            Wdw = P[d.wname + ".weight"].astype(np.float64)[:, 0]      # [G, 3, 3]
            bdw = P[d.wname + ".bias"].astype(np.float64) if d.bias else np.zeros(d.cout)
            if d.post_bn:
                a2, b2 = _bn_affine(P, d.wname + ".post_bn")
                Wdw = Wdw * a2[:, None, None]
                bdw = bdw * a2 + b2
            Wd = np.zeros((9, gp), dtype=np.float32)
            Wd[:, :d.cout] = Wdw.reshape(d.cout, 9).T
            dw_off = blob.add(Wd)[0]
            dwb_off = blob.add(padded(bdw, gp))[0]
            dws_off = blob.add(padded(P[d.wname + ".prelu"], gp))[0] if d.act == "prelu" else -1
            W2, b2 = folded(m)
            w2_off, w2_bytes = blob.add(pack_weights(W2, gp, cout_p))
            b2_off = blob.add(padded(b2, cout_p)[None, :])[0]
            s2_off = blob.add(padded(P[m.wname + ".prelu"], cout_p))[0] if m.act == "prelu" else -1
            _, ho, wo = shp[m.name]
            dst = new_tensor(m.name, m.cout, ho, wo)
            emit(m.name, type=OP_MBBLOCK, src=tid[n.src], dst=dst, res=tid[m.res] if m.res else -1, kh=3, kw=3, stride=d.stride, pad=1, cin=n.cin,
                 cout=m.cout, act=ACT[m.act], flags=0, woff=w2_off, wbytes=w2_bytes, boff=b2_off, soff=s2_off, wrows=cout_p)
            r = ops[-1]
            r[20], r[21], r[22], r[23], r[24], r[25], r[28], r[29], r[30] = w1_off, b1_off, s1_off, ACT[n.act], dw_off, dwb_off, dws_off, ACT[d.act], gp
            _, h1, w1_ = shp[n.name]
            macs = h1 * w1_ * n.cout * n.cin + ho * wo * d.cout * 9 + ho * wo * m.cout * m.cin
            r[26], r[27] = macs & 0xFFFFFFFF, macs >> 32
            if r[26] >= 2 ** 31:
                r[26] -= 2 ** 32
            op_nodes[-1] = [n.name, d.name, m.name]
            out.fused_groups[m.name] = [n.name, d.name, m.name]
            skip.add(ni_ + 1)
            skip.add(ni_ + 2)
        elif n.kind == "conv" and n.groups == 1 and _latfpn(net, ni_, tensors, tid) is not None:
            # A PAFPN level: lateral 1x1 (+ upsampled coarser lateral) and the 3x3 conv on it as ONE launch (csrc/lat_fpn.hip); the lateral is a
            # tensor of its own only when a finer level adds it.  The record is the 3x3 conv's; words 20-22: the lateral's weights / bias / tensor.
            mj, m, keep_lat = _latfpn(net, ni_, tensors, tid)
            src_t = tensors[tid[n.src]]
            W0, b0 = folded(n)
            w0off = blob.add(pack_weights(W0, src_t[1], 64).reshape(64, src_t[1]))[0]
            b0off = blob.add(padded(b0, 64))[0]
            W1, b1 = folded(m)
            w1off, w1bytes = blob.add(repack_kind2(pack_weights(W1, 64, 64)))
            b1off = blob.add(padded(b1, 64)[None, :])[0]
            _, ho, wo = shp[m.name]
            lat_t = new_tensor(n.name, n.cout, ho, wo) + 1 if keep_lat else 0
            dst = new_tensor(m.name, m.cout, ho, wo)
            emit(m.name, type=OP_LATFPN, src=tid[n.src], dst=dst, res=tid[n.res] if n.res else -1, kh=3, kw=3, stride=1, pad=1, cin=m.cin, cout=m.cout,
                 act=0, flags=0, woff=w1off, wbytes=w1bytes, boff=b1off, soff=-1, wrows=64)
            r = ops[-1]
            r[20], r[21], r[22] = w0off, b0off, lat_t
            macs = ho * wo * (n.cout * n.cin + m.cout * m.cin * 9)
            r[26], r[27] = macs & 0xFFFFFFFF, macs >> 32
            if r[26] >= 2 ** 31:
                r[26] -= 2 ** 32
            op_nodes[-1] = [n.name, m.name]
            out.fused_groups[m.name] = [n.name, m.name]
            skip.add(mj)
        elif n.kind == "conv" and n.groups == 1:
            cin, cout = n.cin, n.cout
            _, ho, wo = shp[n.name]
            src_t = tensors[tid[n.src]]
            cin_p, cout_p = src_t[1], _rup(cout, CPAD)
            W, bias_tab, k, stride = fold_conv(n, (src_t[2], src_t[3]))
            pad = n.pad
            ncls = bias_tab.shape[0]
            Wp = pack_weights(W, cin_p, cout_p)
            sc, sc_op = pending_sc.pop(n.name, (None, -1))
            woff, wbytes = blob.add(Wp)
            bt = np.zeros((ncls, cout_p), dtype=np.float32)
            bt[:, :cout] = bias_tab
            boff, _ = blob.add(bt)
            if sc is not None:                                        # second image: weight rows [9 * Cin_p | Cin2_p], one bias row = both biases
                Wsc, bsc = folded(sc)
                if sc.pre_avgpool:                                    # average pool + 1x1 = a 2x2 / stride-2 kernel with W / 4: four taps (dy2, dx2)
                    Wsc = np.repeat(np.repeat(Wsc, 2, axis=2), 2, axis=3) / 4.0
                x_t = tensors[tid[sc.src]]
                assert ncls == 1 and ops[sc_op][2] == tid[sc.name]
                W2p = np.concatenate([Wp.reshape(cout_p, -1), pack_weights(Wsc, x_t[1], cout_p).reshape(cout_p, -1)], axis=1)
                w2off = blob.add(W2p)[0]
                bt2 = np.zeros((1, cout_p), dtype=np.float32)
                bt2[0, :cout] = bias_tab[0] + bsc
                b2off = blob.add(bt2)[0]
            soff = blob.add(padded(P[n.wname + ".prelu"], cout_p))[0] if n.act == "prelu" else -1
            flags = (CF_BORDER if ncls == 9 else 0) | (CF_RES_UP2 if n.res_up2 else 0)
            if n.res_up2:
                # the kernels read the coarser lateral at (y >> 1, x >> 1): nearest 2x.  The reference PAFPN interpolates to the finer level's
                # exact SIZE, which differs for odd or non-2x levels -- unpinned here (the oracle upsamples by scale 2 too), so refused
                r_t = tensors[tid[n.res]]
                if (ho, wo) != (2 * r_t[2], 2 * r_t[3]):
                    raise ValueError(f"{n.name}: top-down add of a {r_t[2]}x{r_t[3]} level onto {ho}x{wo}: only exact 2x levels are supported "
                                     "(input sizes that are multiples of 32)")
            dst = new_tensor(n.name, cout, ho, wo)
            emit(n.name, type=OP_CONV, src=tid[n.src], dst=dst, res=tid[n.res] if n.res else -1, kh=k, kw=k,
                 stride=stride, pad=pad, cin=cin, cout=cout, act=ACT[n.act], flags=flags, woff=woff,
                 wbytes=wbytes, boff=boff, soff=soff, wrows=cout_p)
            if sc is not None:                                        # csrc/net.h W_X_SRC2 ..: block input, taps, row length, stride, second images, the shortcut's op
                r = ops[-1]
                t2, kw2, s2 = (4, 2, 2) if sc.pre_avgpool else (1, 1, sc.stride)
                r[23], r[24], r[25], r[28], r[29], r[30], r[31] = tid[sc.src] + 1, t2, kw2, s2, w2off, b2off, sc_op + 1
                ops[sc_op][29] = len(ops)                             # the shortcut op knows the conv that may absorb it (index + 1)
        elif n.kind == "conv":                                        # depthwise
            assert n.groups == n.cin == n.cout and not n.pre_bn and not n.pre_avgpool and n.res is None
            c = n.cin
            _, ho, wo = shp[n.name]
            cp = tensors[tid[n.src]][1]
            W = P[n.wname + ".weight"].astype(np.float64)[:, 0]       # [c, k, k]
            b = P[n.wname + ".bias"].astype(np.float64) if n.bias else np.zeros(c)
            if n.post_bn:
                a2, b2 = _bn_affine(P, n.wname + ".post_bn")
                W = W * a2[:, None, None]
                b = b * a2 + b2
            Wd = np.zeros((n.k * n.k, cp), dtype=np.float32)
            Wd[:, :c] = W.reshape(c, -1).T
            woff, wbytes = blob.add(Wd)
            boff, _ = blob.add(padded(b, cp))
            soff = blob.add(padded(P[n.wname + ".prelu"], cp))[0] if n.act == "prelu" else -1
            m = nxt
            # (opt-in, FID_DWPW_FUSE=1: measured SLOWER than the two launches -- MobileFaceNet at 32 faces 0.532 vs 0.502 ms, one face 0.350 vs
            # 0.299 ms: these layers are a few hundred 2-us items, two wide launches hide their load latencies better than one deep one)
            if (os.environ.get("FID_DWPW_FUSE") and m is not None and m.kind == "conv" and m.groups == 1 and m.k == 1 and m.pad == 0
                    and m.stride == 1 and m.src == n.name and not m.pre_bn and not m.pre_avgpool and not m.res_up2 and n.k == 3 and n.pad == 1
                    and n.stride in (1, 2) and cp in (32, 64, 128, 256, 512) and n.name not in net.outputs and m.res != n.name
                    and not any(getattr(x, "src", None) == n.name or getattr(x, "res", None) == n.name for x in net.nodes if x is not m)):
                # depthwise 3x3 + the pointwise 1x1 that consumes it: one op, the depthwise result stays in LDS (csrc/dwpw.hip).  The record is
                # the pointwise conv's; words 20-23 hold the depthwise tables / activation, 26 / 27 the MACs of both layers.
                Wm, bm = folded(m)
                cout_p = _rup(m.cout, CPAD)
                pw_off, pw_bytes = blob.add(pack_weights(Wm, cp, cout_p))
                pb_off, _ = blob.add(padded(bm, cout_p)[None, :])
                ps_off = blob.add(padded(P[m.wname + ".prelu"], cout_p))[0] if m.act == "prelu" else -1
                _, mho, mwo = shp[m.name]
                dst = new_tensor(m.name, m.cout, mho, mwo)
                emit(m.name, type=OP_DWPW, src=tid[n.src], dst=dst, res=tid[m.res] if m.res else -1, kh=3, kw=3, stride=n.stride, pad=1, cin=c,
                     cout=m.cout, act=ACT[m.act], flags=0, woff=pw_off, wbytes=pw_bytes, boff=pb_off, soff=ps_off, wrows=cout_p)
                ops[-1][20], ops[-1][21], ops[-1][22], ops[-1][23] = woff, boff, soff, ACT[n.act]
                macs = ho * wo * c * n.k * n.k + mho * mwo * m.cout * m.cin
                ops[-1][26], ops[-1][27] = macs & 0xFFFFFFFF, macs >> 32
                if ops[-1][26] >= 2 ** 31:
                    ops[-1][26] -= 2 ** 32
                op_nodes[-1] = [n.name, m.name]
                out.fused_groups[m.name] = [n.name, m.name]
                skip.add(ni_ + 1)
                continue
            dst = new_tensor(n.name, c, ho, wo, Cp=cp)
            emit(n.name, type=OP_DWCONV, src=tid[n.src], dst=dst, kh=n.k, kw=n.k, stride=n.stride, pad=n.pad,
                 cin=c, cout=c, act=ACT[n.act], woff=woff, wbytes=wbytes, boff=boff, soff=soff, wrows=cp, groups=c)
        elif n.kind == "maxpool":
            c, ho, wo = shp[n.name]
            dst = new_tensor(n.name, c, ho, wo, Cp=tensors[tid[n.src]][1])
            emit(n.name, type=OP_MAXPOOL, src=tid[n.src], dst=dst, kh=n.k, kw=n.k, stride=n.stride, pad=n.pad,
                 cin=c, cout=c)
        elif n.kind == "fc":
            src_t = tensors[tid[n.src]]
            C_, Cp_, H_, W_ = src_t[0], src_t[1], src_t[2], src_t[3]
            assert (C_, H_, W_) == (n.c, n.h, n.w)
            Wm = P[n.wname + ".weight"].astype(np.float64).reshape(n.cout, n.c, n.h, n.w)
            b = P[n.wname + ".bias"].astype(np.float64) if n.bias else np.zeros(n.cout)
            if n.pre_bn:
                a1, b1 = _bn_affine(P, n.wname + ".pre_bn")
                b = b + np.einsum("ochw,c->o", Wm, b1)
                Wm = Wm * a1[None, :, None, None]
            if n.post_bn:
                a2, b2 = _bn_affine(P, n.wname + ".post_bn")
                Wm = Wm * a2[:, None, None, None]
                b = b * a2 + b2
            K = H_ * W_ * Cp_
            cout_p = _rup(n.cout, CPAD)
            Wk = np.zeros((cout_p, H_, W_, Cp_), dtype=np.float32)     # NHWC flatten order of the activation
            Wk[: n.cout, :, :, : n.c] = Wm.transpose(0, 2, 3, 1)
            h = Wk.reshape(cout_p, 1, K).astype(np.float16)
            woff, wbytes = blob.add(h)
            boff, _ = blob.add(padded(b, cout_p)[None, :])
            view = new_tensor(n.name + ".in_view", K, 1, 1, Cp=K)     # alias of the source tensor
            tensors[view][5] = -2 - tid[n.src]                        # resolved to the source's slot below
            dst = new_tensor(n.name, n.cout, 1, 1, dtype=1)
            emit(n.name, type=OP_CONV, src=view, dst=dst, kh=1, kw=1, stride=1, pad=0, cin=n.c * n.h * n.w,
                 cout=n.cout, act=0, flags=0, woff=woff, wbytes=wbytes, boff=boff, wrows=cout_p)
        elif n.kind == "dethead":
            A = n.num_anchors
            _, h_, w_ = shp[n.name]
            src_t = tensors[tid[n.src]]
            s = float(P[n.wname + ".bbox.scale"][0])
            W = np.concatenate([P[n.wname + ".cls.weight"], P[n.wname + ".bbox.weight"] * s,
                                P[n.wname + ".kps.weight"]], axis=0).astype(np.float64)
            b = np.concatenate([P[n.wname + ".cls.bias"], P[n.wname + ".bbox.bias"] * s,
                                P[n.wname + ".kps.bias"]]).astype(np.float64)
            cout = 15 * A
            cout_p = _rup(cout, CPAD)
            Wp = pack_weights(W, src_t[1], cout_p)
            woff, wbytes = blob.add(Wp)
            boff, _ = blob.add(padded(b, cout_p)[None, :])
            dst = new_tensor(n.name, cout, h_, w_, dtype=1)
            emit(n.name, type=OP_CONV, src=tid[n.src], dst=dst, kh=n.k, kw=n.k, stride=1, pad=n.k // 2, cin=n.cin,
                 cout=cout, act=0, flags=0, nsig=A, woff=woff, wbytes=wbytes, boff=boff, wrows=cout_p)
            out.heads[n.name] = dict(stride=n.stride, anchors=A, cp=cout_p, h=h_, w=w_,
                                     score=(0, 1), bbox=(A, 4), kps=(5 * A, 10))
        else:
            raise ValueError(f"cannot lower node {n}")

    # ---- liveness-based slot assignment -------------------------------------------------------
    n_t = len(tensors)
    base = list(range(n_t))                       # alias -> base tensor
    for t in range(n_t):
        if tensors[t][5] <= -2:
            base[t] = -2 - tensors[t][5]
    last_use = [-1] * n_t
    for oi, rec in enumerate(ops):
        for w in (1, 3):
            if rec[w] >= 0:
                last_use[base[rec[w]]] = oi
        if rec[0] == OP_CONV and rec[23] > 0:         # the fused shortcut conv's input (the block input)
            last_use[base[rec[23] - 1]] = oi
    keep = {tid[o] for o in net.outputs}
    slot_of = [-1] * n_t
    slot_size: List[int] = []
    free: List[int] = []

    def nbytes(t):
        return tensors[t][1] * tensors[t][2] * tensors[t][3] * (4 if tensors[t][4] == 1 else 2)

    for oi, rec in enumerate(ops):
        dsts = [rec[2]] + ([rec[20] - 1] if rec[0] == OP_CONV and rec[20] > 0 else [])    # (a fused shortcut + conv op writes two tensors)
        if rec[0] == OP_STEMBLOCK and rec[24] > 0:
            dsts.append(rec[24] - 1)                                                      # (... and so does the fused stem block)
        if rec[0] == OP_LATFPN and rec[22] > 0:
            dsts.append(rec[22] - 1)                                                      # (... and the fused lateral + fpn op when a finer level adds its lateral)
        for d in dsts:
            need = nbytes(d)
            if d in keep or not free:
                slot_size.append(need)
                slot_of[d] = len(slot_size) - 1
            else:
                best = min(free, key=lambda s: (slot_size[s] < need, abs(slot_size[s] - need)))
                free.remove(best)
                slot_size[best] = max(slot_size[best], need)
                slot_of[d] = best
        for t in range(n_t):
            if base[t] == t and slot_of[t] >= 0 and last_use[t] == oi and t not in keep and t not in dsts:
                free.append(slot_of[t])
        for d in dsts:
            if last_use[d] < 0 and d not in keep:      # produced but never read
                free.append(slot_of[d])
    for t in range(n_t):
        tensors[t][5] = slot_of[base[t]]
        if t in keep:
            tensors[t][6] = 1
        assert tensors[t][5] >= 0, t

    out.ops = np.asarray(ops, dtype=np.int32).reshape(-1, OP_WORDS)
    out.tensors = np.asarray(tensors, dtype=np.int32).reshape(-1, TENSOR_WORDS)
    out.blob = blob.bytes()
    out.tensor_id = tid
    out.op_names = op_names
    out.op_nodes = op_nodes
    out.outputs = list(net.outputs)
    out.in_hw = net.in_hw
    return out

"""Network descriptions for the two session.run() calls on the hot path.

The reference never describes its networks: it hands an .onnx file to onnxruntime
(reference models/scrfd.py:59-62,83 and models/arcface.py:18-21,51).  What those files
contain (SCRFD-10G/2.5G/500M, ArcFace IResNet-50 / MobileFaceNet) is restated here as a small
graph IR of *unfused* layers (conv, BN, PReLU, add, pool ...) carrying raw fp32 parameters,
exactly the information an ONNX file holds.  Two consumers read it:

  * lower.py (product): folds BN, packs fp16 NHWC weights and emits the layer table the HIP
    executor runs (csrc/net.cpp);
  * oracle/nets.py (tests only): interprets the same graph layer by layer in fp32 on the CPU.

Node semantics (all tensors NCHW in the description; storage layout is the consumer's business):

  Conv:    y = act( post_bn( conv( avgpool2( pre_bn(x) ) ) + bias ) + up2(res) )
           every stage optional; `act` in {none, relu, prelu}; conv is k x k, stride, pad,
           groups in {1, cin}.
  MaxPool: k x k / stride / pad, floor mode.
  FC:      y = post_bn1d( W . flatten_CHW( pre_bn2d(x) ) + b )
  DetHead: the three SCRFD output convs on one feature map: cls (sigmoid), bbox (x scale), kps;
           returned as [H*W*A, 1|4|10] in (y, x, anchor) order like the ONNX graph does
           (consumed by reference models/scrfd.py:89-119).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

BN_EPS = 1e-5


@dataclass
class Conv:
    name: str
    src: str
    cin: int
    cout: int
    k: int = 3
    stride: int = 1
    pad: int = 1
    wname: Optional[str] = None      # parameter prefix; lets several nodes share weights
    bias: bool = False
    pre_bn: bool = False
    post_bn: bool = True
    act: str = "none"
    res: Optional[str] = None
    res_up2: bool = False
    pre_avgpool: bool = False
    groups: int = 1
    kind: str = "conv"

    def __post_init__(self):
        if self.wname is None:
            self.wname = self.name


@dataclass
class MaxPool:
    name: str
    src: str
    c: int
    k: int = 3
    stride: int = 2
    pad: int = 1
    kind: str = "maxpool"


@dataclass
class FC:
    name: str
    src: str
    c: int
    h: int
    w: int
    cout: int
    bias: bool = True
    pre_bn: bool = True
    post_bn: bool = True
    kind: str = "fc"
    wname: Optional[str] = None

    def __post_init__(self):
        if self.wname is None:
            self.wname = self.name


@dataclass
class DetHead:
    name: str
    src: str
    cin: int
    stride: int
    num_anchors: int = 2
    k: int = 3
    wname: Optional[str] = None
    kind: str = "dethead"

    def __post_init__(self):
        if self.wname is None:
            self.wname = self.name


@dataclass
class Net:
    name: str
    in_hw: Tuple[int, int]
    in_mean: float
    in_scale: float              # blob = (pixel - mean) * scale, BGR->RGB (reference scrfd.py:76-82)
    nodes: List = field(default_factory=list)
    outputs: List[str] = field(default_factory=list)
    kind: str = "net"

    def add(self, node):
        self.nodes.append(node)
        return node.name


# --------------------------------------------------------------------------------------------
# architectures
# --------------------------------------------------------------------------------------------

def _basic_block(net: Net, prefix: str, src: str, cin: int, planes: int, stride: int) -> str:
    """ResNetV1e BasicBlock (SCRFD backbones): conv-bn-relu, conv-bn, + shortcut, relu.
    Shortcut of a strided / widening block is avgpool2 + conv1x1 + bn ("avg_down")."""
    if stride != 1 or cin != planes:
        sc = net.add(Conv(f"{prefix}.down", src, cin, planes, k=1, stride=1, pad=0,
                          pre_avgpool=(stride != 1)))
    else:
        sc = src
    c1 = net.add(Conv(f"{prefix}.conv1", src, cin, planes, k=3, stride=stride, pad=1, act="relu"))
    return net.add(Conv(f"{prefix}.conv2", c1, planes, planes, k=3, stride=1, pad=1, act="relu", res=sc))


def scrfd_resnet(name: str, hw=(640, 640), stem=28, planes=(56, 88, 88, 224), blocks=(3, 4, 2, 3),
                 neck=56, head_ch=80, head_convs=3, head_shared=True) -> Net:
    """SCRFD with a ResNetV1e backbone + PAFPN + stacked-conv head (10G and 2.5G variants)."""
    net = Net(name, hw, 127.5, 1.0 / 128.0)
    x = net.add(Conv("stem.0", "input", 3, stem, k=3, stride=2, pad=1, act="relu"))
    x = net.add(Conv("stem.1", x, stem, stem, act="relu"))
    x = net.add(Conv("stem.2", x, stem, planes[0], act="relu"))
    x = net.add(MaxPool("stem.pool", x, planes[0]))
    cin = planes[0]
    feats = []
    for li, (p, nb) in enumerate(zip(planes, blocks)):
        for bi in range(nb):
            stride = 2 if (bi == 0 and li > 0) else 1
            x = _basic_block(net, f"layer{li + 1}.{bi}", x, cin, p, stride)
            cin = p
        feats.append((x, p))
    c3, c4, c5 = feats[1], feats[2], feats[3]
    # PAFPN (no norm, no activation; biased convs)
    kw = dict(bias=True, post_bn=False)
    lat2 = net.add(Conv("neck.lat2", c5[0], c5[1], neck, k=1, pad=0, **kw))
    lat1 = net.add(Conv("neck.lat1", c4[0], c4[1], neck, k=1, pad=0, res=lat2, res_up2=True, **kw))
    lat0 = net.add(Conv("neck.lat0", c3[0], c3[1], neck, k=1, pad=0, res=lat1, res_up2=True, **kw))
    i0 = net.add(Conv("neck.fpn0", lat0, neck, neck, **kw))
    i1p = net.add(Conv("neck.fpn1", lat1, neck, neck, **kw))
    i2p = net.add(Conv("neck.fpn2", lat2, neck, neck, **kw))
    i1 = net.add(Conv("neck.ds0", i0, neck, neck, stride=2, res=i1p, **kw))
    i2 = net.add(Conv("neck.ds1", i1, neck, neck, stride=2, res=i2p, **kw))
    o1 = net.add(Conv("neck.pa0", i1, neck, neck, **kw))
    o2 = net.add(Conv("neck.pa1", i2, neck, neck, **kw))
    for li, (f, stride) in enumerate(zip((i0, o1, o2), (8, 16, 32))):
        x = f
        c = neck
        for ci in range(head_convs):
            wn = f"head.tower.{ci}" if head_shared else f"head.s{stride}.tower.{ci}"
            x = net.add(Conv(f"head.s{stride}.tower.{ci}", x, c, head_ch, act="relu", wname=wn))
            c = head_ch
        net.add(DetHead(f"head.s{stride}.out", x, head_ch, stride))
    net.outputs = [f"head.s{s}.out" for s in (8, 16, 32)]
    return net


def scrfd_10g(hw=(640, 640)) -> Net:
    """SCRFD-10G (det_10g.onnx, "scrfd_10g_bnkps").  Round 5: per-stride head towers like the other two BN models -- with the
    towers shared across strides (rounds 1-4) the table holds 3.92 M values = 14.97 MiB, 7 % short of the 16.1 MiB file the
    reference's README.md:59 lists; with one tower per stride 4.24 M = 16.17 MiB (+0.4 %).  Same FLOPs, same shapes."""
    return scrfd_resnet("scrfd_10g", hw, head_shared=False)


def scrfd_2_5g(hw=(640, 640)) -> Net:
    return scrfd_resnet("scrfd_2.5g", hw, stem=24, planes=(24, 48, 48, 80), blocks=(3, 5, 3, 2),
                        neck=24, head_ch=64, head_convs=2, head_shared=False)


def _dw_sep(net: Net, prefix: str, src: str, cin: int, cout: int, stride: int) -> str:
    """MobileNet-v1 block: depthwise 3x3 (+bn+relu) then pointwise 1x1 (+bn+relu)."""
    d = net.add(Conv(f"{prefix}.dw", src, cin, cin, k=3, stride=stride, pad=1, groups=cin, act="relu"))
    return net.add(Conv(f"{prefix}.pw", d, cin, cout, k=1, stride=1, pad=0, act="relu"))


def scrfd_500m(hw=(640, 640)) -> Net:
    """SCRFD-500M: MobileNet-v1 style depthwise-separable backbone, PAFPN 16 ch, dw-separable head."""
    net = Net("scrfd_500m", hw, 127.5, 1.0 / 128.0)
    x = net.add(Conv("stem.0", "input", 3, 16, k=3, stride=2, pad=1, act="relu"))
    x = _dw_sep(net, "stem.1", x, 16, 16, 1)
    cin = 16
    feats = []
    for li, (p, nb) in enumerate(zip((40, 72, 152, 288), (2, 3, 2, 6))):
        for bi in range(nb):
            x = _dw_sep(net, f"layer{li + 1}.{bi}", x, cin, p, 2 if bi == 0 else 1)
            cin = p
        feats.append((x, p))
    c3, c4, c5 = feats[1], feats[2], feats[3]
    neck = 16
    kw = dict(bias=True, post_bn=False)
    lat2 = net.add(Conv("neck.lat2", c5[0], c5[1], neck, k=1, pad=0, **kw))
    lat1 = net.add(Conv("neck.lat1", c4[0], c4[1], neck, k=1, pad=0, res=lat2, res_up2=True, **kw))
    lat0 = net.add(Conv("neck.lat0", c3[0], c3[1], neck, k=1, pad=0, res=lat1, res_up2=True, **kw))
    i0 = net.add(Conv("neck.fpn0", lat0, neck, neck, **kw))
    i1p = net.add(Conv("neck.fpn1", lat1, neck, neck, **kw))
    i2p = net.add(Conv("neck.fpn2", lat2, neck, neck, **kw))
    i1 = net.add(Conv("neck.ds0", i0, neck, neck, stride=2, res=i1p, **kw))
    i2 = net.add(Conv("neck.ds1", i1, neck, neck, stride=2, res=i2p, **kw))
    o1 = net.add(Conv("neck.pa0", i1, neck, neck, **kw))
    o2 = net.add(Conv("neck.pa1", i2, neck, neck, **kw))
    for f, stride in zip((i0, o1, o2), (8, 16, 32)):
        x, c = f, neck
        for ci in range(2):
            x = _dw_sep(net, f"head.s{stride}.tower.{ci}", x, c, 64, 1)
            c = 64
        net.add(DetHead(f"head.s{stride}.out", x, 64, stride))
    net.outputs = [f"head.s{s}.out" for s in (8, 16, 32)]
    return net


def iresnet50(hw=(112, 112), layers=(3, 4, 14, 3), name="arcface_r50") -> Net:
    """ArcFace IResNet-50: BN-conv-BN-PReLU-conv-BN + shortcut, no post-add activation."""
    net = Net(name, hw, 127.5, 1.0 / 127.5)
    x = net.add(Conv("stem", "input", 3, 64, k=3, stride=1, pad=1, act="prelu"))
    cin = 64
    for li, (p, nb) in enumerate(zip((64, 128, 256, 512), layers)):
        for bi in range(nb):
            stride = 2 if bi == 0 else 1
            pre = f"layer{li + 1}.{bi}"
            if stride != 1 or cin != p:
                sc = net.add(Conv(f"{pre}.down", x, cin, p, k=1, stride=stride, pad=0))
            else:
                sc = x
            c1 = net.add(Conv(f"{pre}.conv1", x, cin, p, k=3, stride=1, pad=1, pre_bn=True, act="prelu"))
            x = net.add(Conv(f"{pre}.conv2", c1, p, p, k=3, stride=stride, pad=1, res=sc))
            cin = p
    fh, fw = hw[0] // 16, hw[1] // 16
    net.add(FC("fc", x, 512, fh, fw, 512))
    net.outputs = ["fc"]
    return net


def mobilefacenet(hw=(112, 112), blocks=(2, 8, 12, 4)) -> Net:
    """ArcFace MobileFaceNet w600k_mbf (insightface arcface_torch MobileFaceNet with scale = 2: 128 / 128 / 256 / 256
    channels, bottleneck widths 128 / 128 / 256 / 256 and 128 / 256 / 512 in the three stride-2 bottlenecks).

    `blocks` = residual bottlenecks per stage.  Round 5: the only model fact the reference holds is the file size,
    README.md:60 -- w600k_mbf.onnx 12.99 MiB = 3.40 M fp32 values.  The round 1-4 table, blocks (1, 4, 6, 2) (2.08 M values, 7.94 MiB,
    0.43 GMAC -- the figure SURVEY B.5 quotes), is 39 % short of it; blocks (2, 8, 12, 4) holds 3.40 M values = 12.96 MiB (-0.2 %) at
    0.91 GMAC.  The file wins: the default is (2, 8, 12, 4); `mobilefacenet_small` keeps the old table.  With blocks[0] == 1 the first
    stage is a single depthwise conv, otherwise a stack of residual bottlenecks (upstream's `if blocks[0] == 1` branch)."""
    net = Net("arcface_mbf" if tuple(blocks) == (2, 8, 12, 4) else "arcface_mbf_" + "_".join(str(b) for b in blocks), hw, 127.5, 1.0 / 127.5)

    def depthwise(prefix, src, cin, cout, groups, stride, residual):
        a = net.add(Conv(f"{prefix}.pw1", src, cin, groups, k=1, pad=0, act="prelu"))
        b = net.add(Conv(f"{prefix}.dw", a, groups, groups, k=3, stride=stride, pad=1, groups=groups, act="prelu"))
        return net.add(Conv(f"{prefix}.pw2", b, groups, cout, k=1, pad=0, res=src if residual else None))

    x = net.add(Conv("conv1", "input", 3, 128, k=3, stride=2, pad=1, act="prelu"))
    if blocks[0] == 1:
        x = net.add(Conv("conv2_dw", x, 128, 128, k=3, pad=1, groups=128, act="prelu"))
    else:
        for i in range(blocks[0]):
            x = depthwise(f"conv_2.{i}", x, 128, 128, 128, 1, True)
    x = depthwise("conv_23", x, 128, 128, 128, 2, False)
    for i in range(blocks[1]):
        x = depthwise(f"conv_3.{i}", x, 128, 128, 128, 1, True)
    x = depthwise("conv_34", x, 128, 256, 256, 2, False)
    for i in range(blocks[2]):
        x = depthwise(f"conv_4.{i}", x, 256, 256, 256, 1, True)
    x = depthwise("conv_45", x, 256, 256, 512, 2, False)
    for i in range(blocks[3]):
        x = depthwise(f"conv_5.{i}", x, 256, 256, 256, 1, True)
    x = net.add(Conv("conv_6_sep", x, 256, 512, k=1, pad=0, act="prelu"))
    fh, fw = hw[0] // 16, hw[1] // 16
    # GDC: depthwise 7x7 "valid" conv + BN -> 512x1x1, flatten, Linear(no bias), BN1d
    x = net.add(Conv("gdc.dw", x, 512, 512, k=fh, stride=1, pad=0, groups=512))
    net.add(FC("fc", x, 512, 1, 1, 512, bias=False, pre_bn=False, post_bn=True))
    net.outputs = ["fc"]
    return net


def mobilefacenet_small(hw=(112, 112)) -> Net:
    """the rounds 1-4 MobileFaceNet table: blocks (1, 4, 6, 2), 0.43 GMAC (see `mobilefacenet`)"""
    return mobilefacenet(hw, blocks=(1, 4, 6, 2))


ARCHS = {
    "scrfd_10g": scrfd_10g,
    "scrfd_2.5g": scrfd_2_5g,
    "scrfd_500m": scrfd_500m,
    "arcface_r50": iresnet50,
    "arcface_mbf": mobilefacenet,
    "arcface_mbf_small": mobilefacenet_small,
}
# the five files reference download.sh:12-16 fetches, by basename
ONNX_BASENAMES = {
    "det_10g": "scrfd_10g", "det_2.5g": "scrfd_2.5g", "det_500m": "scrfd_500m",
    "w600k_r50": "arcface_r50", "w600k_mbf": "arcface_mbf",
}


# --------------------------------------------------------------------------------------------
# shape inference and cost
# --------------------------------------------------------------------------------------------

def infer_shapes(net: Net) -> Dict[str, Tuple[int, int, int]]:
    """name -> (C, H, W) for every tensor."""
    shp = {"input": (3, net.in_hw[0], net.in_hw[1])}
    for n in net.nodes:
        c, h, w = shp[n.src]
        if n.kind == "conv":
            assert c == n.cin, (n.name, c, n.cin)
            if n.pre_avgpool:
                h, w = h // 2, w // 2
            ho = (h + 2 * n.pad - n.k) // n.stride + 1
            wo = (w + 2 * n.pad - n.k) // n.stride + 1
            shp[n.name] = (n.cout, ho, wo)
        elif n.kind == "maxpool":
            shp[n.name] = (c, (h + 2 * n.pad - n.k) // n.stride + 1, (w + 2 * n.pad - n.k) // n.stride + 1)
        elif n.kind == "fc":
            assert (c, h, w) == (n.c, n.h, n.w), (n.name, (c, h, w))
            shp[n.name] = (n.cout, 1, 1)
        elif n.kind == "dethead":
            shp[n.name] = (n.num_anchors * 15, h, w)
    return shp


def count_macs(net: Net) -> int:
    """Multiply-accumulates of one forward pass (true channel counts, no padding)."""
    shp = infer_shapes(net)
    total = 0
    for n in net.nodes:
        if n.kind == "conv":
            _, ho, wo = shp[n.name]
            total += ho * wo * n.cout * (n.cin // n.groups) * n.k * n.k
        elif n.kind == "fc":
            total += n.c * n.h * n.w * n.cout
        elif n.kind == "dethead":
            _, h, w = shp[n.name]
            total += h * w * n.num_anchors * 15 * n.cin * n.k * n.k
    return total


# --------------------------------------------------------------------------------------------
# synthetic parameters (no trained weights exist offline: reference weights/ is empty)
# --------------------------------------------------------------------------------------------

def _bn(rng, c, gamma=(0.8, 1.2)):
    return {
        "gamma": rng.uniform(gamma[0], gamma[1], c).astype(np.float32),
        "beta": rng.normal(0, 0.05, c).astype(np.float32),
        "mean": rng.normal(0, 0.05, c).astype(np.float32),
        "var": rng.uniform(0.8, 1.25, c).astype(np.float32),
    }


def synth_params(net: Net, seed: int = 0) -> Dict[str, np.ndarray]:
    """He-normal conv weights, near-identity BN statistics, PReLU slope 0.25 (+-), seeded.
    Residual-branch BNs get a small gamma so 24 stacked blocks stay inside fp16 range."""
    rng = np.random.default_rng(seed)
    P: Dict[str, np.ndarray] = {}
    for n in net.nodes:
        wn = getattr(n, "wname", None)
        if n.kind == "conv":
            if wn + ".weight" in P:
                continue
            fan_in = (n.cin // n.groups) * n.k * n.k
            gain = 2.0 if n.act in ("relu", "prelu") else 1.0
            P[wn + ".weight"] = (rng.standard_normal((n.cout, n.cin // n.groups, n.k, n.k))
                                 * np.sqrt(gain / fan_in)).astype(np.float32)
            if n.bias:
                P[wn + ".bias"] = rng.normal(0, 0.05, n.cout).astype(np.float32)
            if n.pre_bn:
                for k, v in _bn(rng, n.cin).items():
                    P[f"{wn}.pre_bn.{k}"] = v
            if n.post_bn:
                g = (0.25, 0.45) if (n.res is not None and not n.res_up2) else (0.8, 1.2)
                for k, v in _bn(rng, n.cout, g).items():
                    P[f"{wn}.post_bn.{k}"] = v
            if n.act == "prelu":
                P[wn + ".prelu"] = rng.uniform(0.15, 0.35, n.cout).astype(np.float32)
        elif n.kind == "fc":
            K = n.c * n.h * n.w
            P[wn + ".weight"] = (rng.standard_normal((n.cout, K)) * np.sqrt(1.0 / K)).astype(np.float32)
            if n.bias:
                P[wn + ".bias"] = rng.normal(0, 0.05, n.cout).astype(np.float32)
            if n.pre_bn:
                for k, v in _bn(rng, n.c).items():
                    P[f"{wn}.pre_bn.{k}"] = v
            if n.post_bn:
                for k, v in _bn(rng, n.cout).items():
                    P[f"{wn}.post_bn.{k}"] = v
        elif n.kind == "dethead":
            A = n.num_anchors
            fan_in = n.cin * n.k * n.k
            std = np.sqrt(1.0 / fan_in)
            P[wn + ".cls.weight"] = (rng.standard_normal((A, n.cin, n.k, n.k)) * std).astype(np.float32)
            # strongly negative prior: few anchors fire, like a trained detector (tuned by calibrate())
            P[wn + ".cls.bias"] = np.full(A, -4.0, dtype=np.float32)
            P[wn + ".bbox.weight"] = (rng.standard_normal((4 * A, n.cin, n.k, n.k)) * std * 0.3).astype(np.float32)
            P[wn + ".bbox.bias"] = rng.uniform(1.5, 3.0, 4 * A).astype(np.float32)   # boxes 3-6 strides wide
            P[wn + ".bbox.scale"] = np.array([rng.uniform(0.9, 1.1)], dtype=np.float32)
            P[wn + ".kps.weight"] = (rng.standard_normal((10 * A, n.cin, n.k, n.k)) * std * 0.15).astype(np.float32)
            # landmark prior: the ArcFace template scaled into the predicted box (a plausible face)
            tmpl = (np.array([[38.2946, 51.6963], [73.5318, 51.5014], [56.0252, 71.7366],
                              [41.5493, 92.3655], [70.7299, 92.2041]]) - 56.0) / 112.0 * 4.5
            P[wn + ".kps.bias"] = np.tile(tmpl.reshape(-1), A).astype(np.float32)
    return P


def params_nbytes(P: Dict[str, np.ndarray]) -> int:
    return int(sum(v.nbytes for v in P.values()))

"""Oracle: the two conv nets behind session.run (reference models/scrfd.py:83, arcface.py:51),
interpreted layer by layer in fp32 with torch-CPU from the unfused graph in
scrfd_arcface_facerecognition_amd/archs.py.  BatchNorm is applied as BatchNorm (not folded), so
this checks the product's folding / packing / fusion as well as its kernels.
Test infrastructure only.  PARITY UNPINNED against onnxruntime (no .onnx, no ORT offline)."""
import numpy as np
import torch
import torch.nn.functional as F

from scrfd_arcface_facerecognition_amd.archs import BN_EPS


def _bn(x, P, prefix):
    g, b = torch.from_numpy(P[prefix + ".gamma"]), torch.from_numpy(P[prefix + ".beta"])
    m, v = torch.from_numpy(P[prefix + ".mean"]), torch.from_numpy(P[prefix + ".var"])
    if x.dim() == 4:
        return F.batch_norm(x, m, v, g, b, False, 0.0, BN_EPS)
    return (x - m) / torch.sqrt(v + BN_EPS) * g + b


@torch.no_grad()
def run_net(net, P, blob, keep=None):
    """blob: float32 [N,3,H,W] (already normalised, RGB).  Returns {name: np.ndarray} for the net
    outputs (plus any tensor named in `keep`).  DetHead outputs are (scores[N,HWA,1],
    bbox[N,HWA,4], kps[N,HWA,10]) like the 9 ONNX outputs of SCRFD."""
    t = {"input": torch.from_numpy(np.ascontiguousarray(blob)).float()}
    for n in net.nodes:
        x = t[n.src]
        if n.kind == "conv":
            w = n.wname
            if n.pre_bn:
                x = _bn(x, P, w + ".pre_bn")
            if n.pre_avgpool:
                x = F.avg_pool2d(x, 2, 2)
            b = torch.from_numpy(P[w + ".bias"]) if n.bias else None
            y = F.conv2d(x, torch.from_numpy(P[w + ".weight"]), b, n.stride, n.pad, 1, n.groups)
            if n.post_bn:
                y = _bn(y, P, w + ".post_bn")
            if n.res is not None:
                r = t[n.res]
                if n.res_up2:
                    r = F.interpolate(r, scale_factor=2, mode="nearest")
                y = y + r
            if n.act == "relu":
                y = F.relu(y)
            elif n.act == "prelu":
                y = F.prelu(y, torch.from_numpy(P[w + ".prelu"]))
            t[n.name] = y
        elif n.kind == "maxpool":
            t[n.name] = F.max_pool2d(x, n.k, n.stride, n.pad)
        elif n.kind == "fc":
            w = n.wname
            if n.pre_bn:
                x = _bn(x, P, w + ".pre_bn")
            y = x.flatten(1) @ torch.from_numpy(P[w + ".weight"]).T
            if n.bias:
                y = y + torch.from_numpy(P[w + ".bias"])
            if n.post_bn:
                y = _bn(y, P, w + ".post_bn")
            t[n.name] = y
        elif n.kind == "dethead":
            w, A = n.wname, n.num_anchors
            pad = n.k // 2
            cls = torch.sigmoid(F.conv2d(x, torch.from_numpy(P[w + ".cls.weight"]),
                                         torch.from_numpy(P[w + ".cls.bias"]), 1, pad))
            bb = F.conv2d(x, torch.from_numpy(P[w + ".bbox.weight"]),
                          torch.from_numpy(P[w + ".bbox.bias"]), 1, pad) * float(P[w + ".bbox.scale"][0])
            kp = F.conv2d(x, torch.from_numpy(P[w + ".kps.weight"]),
                          torch.from_numpy(P[w + ".kps.bias"]), 1, pad)
            N = x.shape[0]
            t[n.name] = (cls.permute(0, 2, 3, 1).reshape(N, -1, 1),
                         bb.permute(0, 2, 3, 1).reshape(N, -1, 4),
                         kp.permute(0, 2, 3, 1).reshape(N, -1, 10))
        else:
            raise ValueError(n.kind)
    names = list(net.outputs) + list(keep or [])
    out = {}
    for k in names:
        v = t[k]
        out[k] = tuple(a.numpy() for a in v) if isinstance(v, tuple) else v.numpy()
    return out


def scrfd_session_outputs(net, P, blob):
    """The 9 arrays SCRFD's session.run returns for ONE image (scrfd.py:83-94 order):
    scores(8,16,32), bbox(8,16,32), kps(8,16,32)."""
    o = run_net(net, P, blob)
    heads = [o[k] for k in net.outputs]
    return ([h[0][0] for h in heads] + [h[1][0] for h in heads] + [h[2][0] for h in heads])

"""A session-like object over a CompiledNet: the seam the reference plugs onnxruntime into
(`onnxruntime.InferenceSession(path, providers=...)`, `.get_inputs()`, `.get_outputs()`, `.run()`;
reference models/scrfd.py:59-65,83 and models/arcface.py:11-37,51 -- ArcFace accepts an injected
`session=`).  `HipSession` offers the same four calls backed by libfaceid, plus `run_images`, the
fast path the mirrored model classes use (uint8 frames straight to the device: no float blob)."""
from __future__ import annotations

from typing import Dict, List

import numpy as np

from ._lib import Context, default_context
from .engine import CompiledNet, resolve_model


class NodeArg:
    def __init__(self, name, shape, type_="tensor(float)"):
        self.name, self.shape, self.type = name, shape, type_


class HipSession:
    def __init__(self, model_path: str, providers=None, *, ctx: Context = None, device: int = 0, input_hw=None,
                 max_batch: int = 8, params=None, net=None):
        self.ctx = ctx or default_context(device)
        if net is None:
            net, params = resolve_model(model_path, input_hw)
        self.net, self.params = net, params
        self.max_batch = int(max_batch)
        self._compiled: Dict[tuple, CompiledNet] = {}
        self.is_detector = net.nodes[-1].kind == "dethead"
        H, W = net.in_hw
        self._inputs = [NodeArg("input.1", [1 if self.is_detector else "None", 3, H if not self.is_detector else "?",
                                            W if not self.is_detector else "?"])]
        if self.is_detector:
            self._outputs = [NodeArg(f"{kind}_{s}", ["?", c]) for kind, c in (("score", 1), ("bbox", 4), ("kps", 10))
                             for s in (8, 16, 32)]
        else:
            self._outputs = [NodeArg("embedding", [1, 512])]

    # -- onnxruntime-like surface -----------------------------------------------------------
    def get_inputs(self) -> List[NodeArg]:
        return self._inputs

    def get_outputs(self) -> List[NodeArg]:
        return self._outputs

    def get_providers(self):
        return ["MI355XExecutionProvider"]

    def compiled(self, hw=None) -> CompiledNet:
        hw = tuple(hw or self.net.in_hw)
        if hw not in self._compiled:
            net = self.net
            if hw != tuple(net.in_hw):
                if not self.is_detector:
                    raise ValueError(f"recognition net is fixed at {net.in_hw}, got {hw}")
                from .archs import ARCHS
                net = ARCHS[self.net.name](hw)
            self._compiled[hw] = CompiledNet(self.ctx, net, self.params, self.max_batch)
        return self._compiled[hw]

    def blob_to_images(self, blob: np.ndarray) -> np.ndarray:
        """Invert cv2.dnn.blobFromImage(s): float32 RGB NCHW -> uint8 BGR NHWC.  The blob must come
        from uint8 pixels (it always does on the reference path); anything else is rejected."""
        blob = np.asarray(blob, dtype=np.float32)
        px = blob / np.float32(self.net.in_scale) + np.float32(self.net.in_mean)
        r = np.rint(px)
        if np.abs(px - r).max() > 1e-2 or r.min() < 0 or r.max() > 255:
            raise ValueError("HipSession.run expects a blob made from uint8 pixels by blobFromImage(s)")
        return np.ascontiguousarray(r.astype(np.uint8).transpose(0, 2, 3, 1)[..., ::-1])

    def run(self, output_names, input_feed):
        (blob,) = input_feed.values()
        outs = self.run_images(self.blob_to_images(blob))
        names = [o.name for o in self._outputs]
        if output_names is None:
            return outs
        return [outs[names.index(n)] for n in output_names]

    # -- native surface -----------------------------------------------------------------------
    def run_images(self, images: np.ndarray) -> List[np.ndarray]:
        """uint8 BGR [B,H,W,3] -> the net's outputs as host arrays in the ONNX layout."""
        images = np.ascontiguousarray(images, dtype=np.uint8)
        B = images.shape[0]
        outs = None
        for b0 in range(0, B, self.max_batch):
            chunk = images[b0:b0 + self.max_batch]
            cn = self.compiled(chunk.shape[1:3])
            cn.run(chunk)
            if self.is_detector:
                if chunk.shape[0] != 1 or B != 1:
                    raise ValueError("the ONNX-layout outputs of the detector are defined for one image "
                                     "(reference scrfd.py:83); use SCRFD.detect_batch for batches")
                parts = {"score": [], "bbox": [], "kps": []}
                for name in cn.low.outputs:
                    h = cn.low.heads[name]
                    fused = cn.read(name, 1)[0]                       # [H, W, 15*A]
                    A = h["anchors"]
                    for key in ("score", "bbox", "kps"):
                        off, c = h[key]
                        parts[key].append(np.ascontiguousarray(fused[..., off:off + A * c]).reshape(-1, c))
                outs = parts["score"] + parts["bbox"] + parts["kps"]
            else:
                e = cn.read(cn.low.outputs[0], chunk.shape[0]).reshape(chunk.shape[0], -1)
                outs = [e] if outs is None else [np.concatenate([outs[0], e])]
        return outs

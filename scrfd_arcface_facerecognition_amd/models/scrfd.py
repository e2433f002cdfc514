"""SCRFD detector with the call surface of reference models/scrfd.py (class SCRFD, :12-207):
same constructor arguments and defaults, same attributes, same `forward` / `detect` / `nms`
results -- computed by libfaceid on an MI355X instead of onnxruntime + numpy."""
from __future__ import annotations

import ctypes as C
from typing import List, Tuple

import numpy as np

from .. import _lib
from .._lib import check
from ..engine import HeadViews, PostProcessor
from ..session import HipSession

__all__ = ["SCRFD"]


class SCRFD:
    def __init__(self, model_path: str, input_size: Tuple[int] = (640, 640), conf_thres: float = 0.5,
                 iou_thres: float = 0.4, *, device: int = 0, ctx=None, max_batch: int = 8, max_det: int = 512,
                 session=None) -> None:
        self.input_size = input_size
        self.conf_thres = conf_thres
        self.iou_thres = iou_thres
        # SCRFD model params (reference scrfd.py:39-47)
        self.fmc = 3
        self._feat_stride_fpn = [8, 16, 32]
        self._num_anchors = 2
        self.use_kps = True
        self.mean = 127.5
        self.std = 128.0
        self.center_cache = {}
        self._device, self._ctx, self._max_batch, self._max_det = device, ctx, int(max_batch), int(max_det)
        self._post = None
        self._session = session        # keyword-only extension: a ready HipSession (like ArcFace(session=...), arcface.py:11-21)
        self._initialize_model(model_path=model_path)

    def _initialize_model(self, model_path: str):
        try:
            self.session = self._session or HipSession(model_path, ctx=self._ctx, device=self._device,
                                                       input_hw=(self.input_size[1], self.input_size[0]),
                                                       max_batch=self._max_batch)
            self.output_names = [x.name for x in self.session.get_outputs()]
            self.input_names = [x.name for x in self.session.get_inputs()]
        except Exception as e:                       # reference scrfd.py:66-68
            print(f"Failed to load the model: {e}")
            raise
        self.ctx = self.session.ctx

    # ------------------------------------------------------------------------------------
    def _postprocessor(self) -> PostProcessor:
        if self._post is None:
            self._post = PostProcessor(self.ctx, self._max_batch, cap=self._max_det)
        return self._post

    def _run_net(self, images_dev, B, hw):
        cn = self.session.compiled(hw)
        cn.run_device(images_dev, B)
        return cn

    def forward(self, image, threshold):
        """scrfd.py:70-120: (scores_list, bboxes_list, kpss_list) per stride, candidates >= threshold in
        anchor order, coordinates in the (letterboxed) input image."""
        image = np.ascontiguousarray(image, dtype=np.uint8)
        H, W = image.shape[:2]
        cn = self.session.compiled((H, W))
        cn.run(image[None])
        hv = HeadViews.from_fused(cn)
        post = self._postprocessor()
        rec = self.ctx.empty((1, post.cand_cap, 16), np.float32)
        cnt = self.ctx.empty((1,), np.int32)
        check(self.ctx.lib.fid_scrfd_decode(self.ctx.handle, C.cast(hv.ptrs, _lib.c_void_pp), hv.pix, hv.anc, hv.bstride,
                                            1, H, W, self._num_anchors, float(threshold), C.c_void_p(rec.ptr),
                                            C.c_void_p(cnt.ptr)))
        n = int(cnt.download()[0])
        if n > post.cand_cap:
            raise _lib.FaceIdError(-3, f"{n} candidates exceed the workspace ({post.cand_cap})")
        r = rec.download()[0, :n]
        flat = r[:, 15].view(np.int32)
        bounds = np.cumsum([(H // s) * (W // s) * self._num_anchors for s in self._feat_stride_fpn])
        level = np.searchsorted(bounds, flat, side="right")
        scores_list, bboxes_list, kpss_list = [], [], []
        for lv in range(self.fmc):
            m = level == lv
            scores_list.append(np.ascontiguousarray(r[m, 4:5]))
            bboxes_list.append(np.ascontiguousarray(r[m, 0:4]))
            kpss_list.append(np.ascontiguousarray(r[m, 5:15]).reshape(-1, 5, 2))
        return scores_list, bboxes_list, kpss_list

    def detect_batch(self, images, max_num=0, metric="max") -> List[Tuple[np.ndarray, np.ndarray]]:
        """Batched detect(): images uint8 [B,H,W,3] (one shape) -> [(det[K,5], kpss[K,5,2])] per frame."""
        images = np.ascontiguousarray(images, dtype=np.uint8)
        assert images.ndim == 4 and images.shape[3] == 3
        results = []
        for b0 in range(0, images.shape[0], self._max_batch):
            chunk = images[b0:b0 + self._max_batch]
            results += self._detect_chunk(chunk, max_num, metric)
        return results

    def _detect_chunk(self, images, max_num, metric):
        B, H, W, _ = images.shape
        in_w, in_h = self.input_size
        frames = self.ctx.to_device(images)
        if (H, W) == (in_h, in_w):
            det_in = frames
        else:                                      # scrfd.py:123-138 on the device
            det_in = self.ctx.empty((B, in_h, in_w, 3), np.uint8)
            sc = C.c_double()
            check(self.ctx.lib.fid_letterbox(self.ctx.handle, C.c_void_p(frames.ptr), B, H, W, C.c_void_p(det_in.ptr),
                                             in_h, in_w, C.byref(sc)))
        cn = self._run_net(det_in, B, (in_h, in_w))
        post = self._postprocessor()
        post.run(HeadViews.from_fused(cn), B, (in_h, in_w), (H, W), self.conf_thres, self.iou_thres, max_num,
                 0 if metric == "max" else 1, self._num_anchors)
        return post.fetch(B)

    def detect(self, image, max_num=0, metric="max"):
        """scrfd.py:122-178: (det float32 [K,5], kpss float32 [K,5,2]); K may be 0."""
        (det, kpss), = self._detect_chunk(np.ascontiguousarray(image, dtype=np.uint8)[None], max_num, metric)
        return det, kpss

    def nms(self, dets, iou_thres):
        """scrfd.py:180-207: indices of the kept rows of dets [K,5], best score first."""
        dets = np.ascontiguousarray(dets, dtype=np.float32)
        K = dets.shape[0]
        if K == 0:
            return []
        d = self.ctx.to_device(dets)
        keep = self.ctx.empty((K,), np.int32)
        cnt = self.ctx.empty((1,), np.int32)
        check(self.ctx.lib.fid_nms(self.ctx.handle, C.c_void_p(d.ptr), K, float(iou_thres), C.c_void_p(keep.ptr),
                                   C.c_void_p(cnt.ptr)))
        n = int(cnt.download()[0])
        return [np.int64(i) for i in keep.download()[:n]]

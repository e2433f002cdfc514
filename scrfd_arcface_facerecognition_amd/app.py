"""`FaceAnalysis`-style front end: what the reference's product layer gets from
`insightface.app.FaceAnalysis(...).get(image)` (smart_face_recognition.py:356-358,1473-1519,
compare_face_from_api.py:69-70,157-174): a list of faces, each with `bbox`, `kps`, `det_score`,
`embedding` and `normed_embedding`, plus the best-face selection by detector score.
All faces of an image are aligned and embedded in ONE batch on the device."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import numpy as np

from ._lib import check
from .models import ArcFace, SCRFD


class Face(dict):
    """dict with attribute access, like insightface's Face"""
    __getattr__ = dict.get

    @property
    def embedding_norm(self):
        return float(np.linalg.norm(self["embedding"])) if self.get("embedding") is not None else None


class FaceAnalysis:
    def __init__(self, det_model: str = "synthetic:scrfd_10g", rec_model: str = "synthetic:arcface_r50", *, device: int = 0,
                 det_size=(640, 640), det_thresh: float = 0.5, max_faces: int = 64):
        self.det = SCRFD(det_model, input_size=det_size, conf_thres=det_thresh, device=device)
        self.rec = ArcFace(rec_model, device=device, ctx=self.det.ctx, max_batch=max_faces)
        self.ctx = self.det.ctx
        self.max_faces = int(max_faces)

    def prepare(self, ctx_id: int = 0, det_size=(640, 640), det_thresh: Optional[float] = None):
        """insightface API compatibility (smart_face_recognition.py:358)"""
        self.det.input_size = det_size
        if det_thresh is not None:
            self.det.conf_thres = det_thresh

    def get(self, image: np.ndarray, max_num: int = 0) -> List[Face]:
        det, kpss = self.det.detect(image, max_num=max_num)
        n = min(len(det), self.max_faces)
        if n == 0:
            return []
        ctx = self.ctx
        H, W = image.shape[:2]
        fr = ctx.to_device(np.ascontiguousarray(image, dtype=np.uint8)[None])
        kp = ctx.to_device(np.ascontiguousarray(kpss[:n], dtype=np.float32).reshape(1, n, 10))
        cn = ctx.to_device(np.array([n], np.int32))
        crops = ctx.empty((n, 112, 112, 3), np.uint8)
        check(ctx.lib.fid_align_crops(ctx.handle, C.c_void_p(fr.ptr), 1, H, W, C.c_void_p(kp.ptr), C.c_void_p(cn.ptr), n, n,
                                      C.c_void_p(crops.ptr), None))
        net = self.rec.session.compiled()
        net.run_device(crops, n)
        emb_ptr, _, _ = net.tensor(net.low.outputs[0])
        q = ctx.empty((n, 512), np.float16)
        check(ctx.lib.fid_l2_normalize_f16(ctx.handle, C.c_void_p(emb_ptr), n, 512, C.c_void_p(q.ptr)))
        emb = net.read(net.low.outputs[0], n).reshape(n, 512)
        normed = q.download().astype(np.float32)
        return [Face(bbox=det[i, :4].copy(), det_score=float(det[i, 4]), kps=kpss[i].copy(), embedding=emb[i].copy(),
                     normed_embedding=normed[i].copy()) for i in range(n)]

    def best_face(self, image: np.ndarray) -> Optional[Face]:
        """the highest-det_score face (smart_face_recognition.py:1480-1492)"""
        faces = self.get(image)
        return max(faces, key=lambda f: f.det_score) if faces else None

"""`FaceAnalysis`-style front end: what the reference's product layer gets from
`insightface.app.FaceAnalysis(...).get(image)` (smart_face_recognition.py:356-358,1473-1519,
compare_face_from_api.py:69-70,157-174): a list of faces, each with `bbox`, `kps`, `det_score`,
`embedding` and `normed_embedding`, plus the product layer's quality scores, side-face test and best-face
selection with its rejections (smart_face_recognition.py:1145-1216,1218-1297,1299-1399) -- `fid_face_gates`.
All faces of an image are aligned and embedded in ONE batch on the device."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import numpy as np

from ._lib import GateConfig, check
from .models import ArcFace, SCRFD

VERDICTS = ("accepted", "no face", "confidence too low", "side face", "quality too low")      # FID_GATE_* (include/faceid.h)
QUALITY_KEYS = ("overall", "blur", "pose", "lighting", "size")


def face_gates(ctx, det_dev, kps_dev, counts_dev, batch: int, cap: int, faces_per_frame: int, config: Optional[GateConfig] = None, pose=None):
    """The reference's quality / side-face gates and best-face selection (smart_face_recognition.py:1145-1216, 1218-1297, 1299-1399,
    1473-1519) for every face of a batch, on the post-process's device arrays (`PostProcessor.det / .kps / .counts`, cap = their
    second dimension).  pose: optional [batch, faces_per_frame, 2] yaw / pitch in radians (0 = not available), handed over as float64.
    -> quality [batch, F, 5] (QUALITY_KEYS), side score [batch, F], side flag [batch, F] (bool), best [batch, 2] = (face index or -1, verdict)"""
    cfg = config or GateConfig()
    F = faces_per_frame
    quality = ctx.empty((batch, F, 5), np.float32)
    side = ctx.empty((batch, F), np.int32)
    best = ctx.empty((batch, 2), np.int32)
    pose_dev = ctx.to_device(np.ascontiguousarray(pose, dtype=np.float64).reshape(batch, F, 2)) if pose is not None else None   # (float64 across the boundary: python floats in the reference)
    check(ctx.lib.fid_face_gates(ctx.handle, C.c_void_p(det_dev.ptr), C.c_void_p(kps_dev.ptr), C.c_void_p(counts_dev.ptr), batch, cap, F,
                                 C.c_void_p(pose_dev.ptr) if pose_dev is not None else None, C.byref(cfg), C.c_void_p(quality.ptr),
                                 C.c_void_p(side.ptr), C.c_void_p(best.ptr)))
    sd = side.download()
    return quality.download(), sd & 0xFFFF, (sd >> 16).astype(bool), best.download()


class Face(dict):
    """dict with attribute access, like insightface's Face"""
    __getattr__ = dict.get

    @property
    def embedding_norm(self):
        return float(np.linalg.norm(self["embedding"])) if self.get("embedding") is not None else None


class FaceAnalysis:
    def __init__(self, det_model: str = "synthetic:scrfd_10g", rec_model: str = "synthetic:arcface_r50", *, device: int = 0,
                 det_size=(640, 640), det_thresh: float = 0.5, max_faces: int = 64, gate_config: Optional[GateConfig] = None):
        self.det = SCRFD(det_model, input_size=det_size, conf_thres=det_thresh, device=device)
        self.rec = ArcFace(rec_model, device=device, ctx=self.det.ctx, max_batch=max_faces)
        self.ctx = self.det.ctx
        self.max_faces = int(max_faces)
        self.gate_config = gate_config or GateConfig()       # the reference's config.json thresholds (GateConfig.from_reference_json)
        self.last_verdict = None                              # of the last best_face(): one of VERDICTS

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
        # quality scores, side-face flag and the best-face verdict of the reference's product layer, all faces in one launch
        dd = ctx.to_device(np.ascontiguousarray(det[:n], dtype=np.float32).reshape(1, n, 5))
        quality, side_score, side_flag, best = face_gates(ctx, dd, kp, cn, 1, n, n, self.gate_config)
        self._best = (int(best[0, 0]), int(best[0, 1]))
        return [Face(bbox=det[i, :4].copy(), det_score=float(det[i, 4]), kps=kpss[i].copy(), embedding=emb[i].copy(),
                     normed_embedding=normed[i].copy(), quality={k: float(quality[0, i, j]) for j, k in enumerate(QUALITY_KEYS)},
                     is_side_face=bool(side_flag[0, i]), side_face_score=int(side_score[0, i])) for i in range(n)]

    def best_face(self, image: np.ndarray) -> Optional[Face]:
        """The reference's enrolment gate (smart_face_recognition.py:1473-1519): the first highest-det_score face, rejected (None, with
        `last_verdict` naming the reason) when its score is below `confidence_threshold`, when it is a side face, or when its overall
        quality is below `min_quality_threshold` -- decided on the device by fid_face_gates."""
        self._best = (-1, 1)
        faces = self.get(image)
        idx, verdict = self._best
        self.last_verdict = VERDICTS[verdict]
        return faces[idx] if verdict == 0 and faces else None

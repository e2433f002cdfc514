"""ArcFace recogniser with the call surface of reference models/arcface.py (class ArcFace, :10-57):
`ArcFace(model_path=None, session=None)`, `.get_feat(images)`, `.__call__(image, kps)` and the same
attributes.  Alignment + embedding run in libfaceid on an MI355X."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .._lib import check
from ..session import HipSession

__all__ = ["ArcFace"]


class ArcFace:
    def __init__(self, model_path: str = None, session=None, *, device: int = 0, ctx=None, max_batch: int = 64) -> None:
        self.session = session
        self.input_mean = 127.5
        self.input_std = 127.5
        self.taskname = "recognition"
        if session is None:
            self.session = HipSession(model_path, ctx=ctx, device=device, max_batch=max_batch)
        input_cfg = self.session.get_inputs()[0]
        input_shape = input_cfg.shape
        self.input_size = tuple(input_shape[2:4][::-1])
        self.input_shape = input_shape
        outputs = self.session.get_outputs()
        self.input_name = input_cfg.name
        self.output_names = [o.name for o in outputs]
        assert len(self.output_names) == 1
        self.output_shape = outputs[0].shape
        self._native = isinstance(self.session, HipSession)
        self.ctx = self.session.ctx if self._native else ctx

    def get_feat(self, images) -> np.ndarray:
        """arcface.py:39-52: one aligned crop or a list (or [N,112,112,3] array) -> float32 [N,512],
        raw (not L2-normalised) embeddings."""
        if isinstance(images, np.ndarray) and images.ndim == 3:
            images = [images]
        imgs = np.ascontiguousarray(np.stack([np.asarray(i, dtype=np.uint8) for i in images]))
        if self._native:
            return self.session.run_images(imgs)[0]
        # an injected foreign session (arcface.py:11-21): hand it the blob it expects
        blob = np.ascontiguousarray(((imgs[..., ::-1].astype(np.float32) - np.float32(self.input_mean))
                                     * np.float32(1.0 / self.input_std)).transpose(0, 3, 1, 2))
        return self.session.run(self.output_names, {self.input_name: blob})[0]

    def align(self, image, kps) -> np.ndarray:
        from ..utils.helpers import norm_crop_image
        return norm_crop_image(image, landmark=kps, ctx=self.ctx)

    def __call__(self, image, kps):
        """arcface.py:54-57: frame + 5 landmarks -> float32 [512]"""
        aligned = self.align(image, kps)
        return self.get_feat(aligned).flatten()

    @staticmethod
    def compute_sim(feat1, feat2) -> np.float32:
        """Alias named by BASELINE.json's north_star; the reference function is
        utils.helpers.compute_similarity (utils/helpers.py:110-123)."""
        from ..utils.helpers import compute_similarity
        return compute_similarity(feat1, feat2)

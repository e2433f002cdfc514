"""Batched det -> align -> embed -> match harness: the logic of reference main.py `build_targets`
(:78-105) and `frame_processor` (:108-150) for a whole batch of frames, with every stage on the
device and no host round trip between stages.

Per step (one batch of B frames, F = max faces kept per frame = the reference's --max-num):

  frames u8 [B,H,W,3] --(letterbox if needed)--> SCRFD net --> post-process (top-F per frame)
       --> align (Umeyama + warp, B*F crops) --> ArcFace net --> L2-normalise (fp16 [B*F,512])
       --> [multi-GPU: ONE all-gather of the per-rank unit embeddings over RCCL/xGMI]
       --> cosine GEMM + arg-max against the gallery  --> (index, score) per face slot

Multi-GPU: frames shard by rank (independent units), weights and gallery are replicated, the only
collective is the embedding all-gather (BASELINE.json north_star).  `FacePipeline` itself is
single-device; `run_step_distributed` adds the collective through torch.distributed
(backend "nccl" = RCCL on ROCm, "gloo" in the CPU tests)."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import Context, check
from .engine import CompiledNet, Gallery, HeadViews, PostProcessor


def shard_range(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block partition of frames over ranks: rank r gets
    [r*n/world, (r+1)*n/world)."""
    return (rank * n_items) // world, ((rank + 1) * n_items) // world


class FacePipeline:
    def __init__(self, ctx: Context, det: CompiledNet, rec: CompiledNet, *, batch: int, faces_per_frame: int = 1,
                 conf_thres: float = 0.5, iou_thres: float = 0.4, metric: int = 0, det_cap: int = 256,
                 q_buffer=None):
        assert rec.max_batch >= batch * faces_per_frame and det.max_batch >= batch
        self.ctx, self.det, self.rec = ctx, det, rec
        self.B, self.F = int(batch), int(faces_per_frame)
        self.conf, self.iou, self.metric = float(conf_thres), float(iou_thres), int(metric)
        self.in_hw = det.in_hw
        self.post = PostProcessor(ctx, batch, cap=det_cap)
        self.n_slots = self.B * self.F
        self.crops = ctx.empty((self.n_slots, 112, 112, 3), np.uint8)
        self.emb_dim = 512
        # unit embeddings: caller may supply the device buffer (e.g. a torch tensor for the all-gather)
        self.q = q_buffer if q_buffer is not None else ctx.empty((self.n_slots, self.emb_dim), np.float16)
        self.idx = ctx.empty((self.n_slots,), np.int32)
        self.score = ctx.empty((self.n_slots,), np.float32)
        self._det_in = None
        self._hv: Optional[HeadViews] = None

    # -- stages ---------------------------------------------------------------------------------
    def detect(self, frames_dev, H, W):
        in_h, in_w = self.in_hw
        if (H, W) == (in_h, in_w):
            det_in = frames_dev
        else:
            if self._det_in is None:
                self._det_in = self.ctx.empty((self.B, in_h, in_w, 3), np.uint8)
            sc = C.c_double()
            check(self.ctx.lib.fid_letterbox(self.ctx.handle, _lib._ptr(frames_dev), self.B, H, W,
                                             C.c_void_p(self._det_in.ptr), in_h, in_w, C.byref(sc)))
            det_in = self._det_in
        self.det.run_device(det_in, self.B)
        if self._hv is None:
            self._hv = HeadViews.from_fused(self.det)
        self.post.run(self._hv, self.B, self.in_hw, (H, W), self.conf, self.iou, self.F, self.metric)

    def embed(self, frames_dev, H, W):
        check(self.ctx.lib.fid_align_crops(self.ctx.handle, _lib._ptr(frames_dev), self.B, H, W,
                                           C.c_void_p(self.post.kps.ptr), C.c_void_p(self.post.counts.ptr),
                                           self.post.cap, self.F, C.c_void_p(self.crops.ptr), None))
        self.rec.run_device(self.crops, self.n_slots)
        emb_ptr, _, _ = self.rec.tensor(self.rec.low.outputs[0])
        check(self.ctx.lib.fid_l2_normalize_f16(self.ctx.handle, C.c_void_p(emb_ptr), self.n_slots, self.emb_dim,
                                                _lib._ptr(self.q)))

    def match(self, gallery: Gallery, thresh: float, q=None, n=None, idx=None, score=None):
        gallery.match_device(self.q if q is None else q, self.n_slots if n is None else n, thresh,
                             self.idx if idx is None else idx, self.score if score is None else score)

    def run_step(self, frames_dev, H, W, gallery: Gallery, thresh: float = 0.4):
        """One full pass over one batch; asynchronous (results stay on the device)."""
        self.detect(frames_dev, H, W)
        self.embed(frames_dev, H, W)
        self.match(gallery, thresh)

    # -- results -----------------------------------------------------------------------------------
    def results(self, gallery: Gallery):
        """Host view of the last step: per frame a list of (bbox[4], det_score, kps[5,2], name, similarity)
        -- what frame_processor draws (main.py:132-148)."""
        self.post.check()
        counts = self.post.counts.download()
        det = self.post.det.download()
        kps = self.post.kps.download()
        idx = self.idx.download().reshape(self.B, self.F)
        score = self.score.download().reshape(self.B, self.F)
        out = []
        for b in range(self.B):
            faces = []
            for f in range(min(int(counts[b]), self.F)):
                j = int(idx[b, f])
                faces.append((det[b, f, :4].copy(), float(det[b, f, 4]), kps[b, f].reshape(5, 2).copy(),
                              gallery.names[j] if j >= 0 else "Unknown", float(score[b, f])))
            out.append(faces)
        return out

    def embeddings(self) -> np.ndarray:
        """Raw (un-normalised) fp32 embeddings of the last step, [B*F, 512]."""
        return self.rec.read(self.rec.low.outputs[0], self.n_slots).reshape(self.n_slots, -1)


def calibrate_detector_bias(ctx: Context, net, params, frames: np.ndarray, target: int = 48, max_batch: int = 8):
    """Synthetic (random-init) detectors fire on ~half of all anchors.  Shift the shared cls bias so that
    the busiest-to-quietest calibration frame keeps at least `target` anchors >= 0.5 -- the job training
    does for a real detector.  Returns a NEW params dict; it feeds the HIP engine and the oracle alike."""
    cn = CompiledNet(ctx, net, params, max_batch=max_batch)
    kth = []
    for b0 in range(0, len(frames), max_batch):
        chunk = frames[b0:b0 + max_batch]
        cn.run(chunk)
        sc = np.concatenate([cn.read(name, len(chunk))[..., :2].reshape(len(chunk), -1) for name in cn.low.outputs], axis=1)
        sc = np.clip(sc.astype(np.float64), 1e-7, 1 - 1e-7)
        logit = np.log(sc / (1 - sc))
        kth += [np.sort(l)[-target] for l in logit]
    cn.close()
    shift = -float(min(kth))
    out = dict(params)
    for k in params:
        if k.endswith(".cls.bias"):
            out[k] = (params[k] + np.float32(shift)).astype(np.float32)
    return out, shift


# ---- multi-GPU step: the single collective -----------------------------------------------------

def run_step_distributed(pipe: FacePipeline, frames_dev, H, W, gallery: Gallery, thresh, q_local, q_all, dist):
    """Local detect/align/embed on this rank's frames, ONE all-gather of the unit embeddings, then this
    rank matches its own block of the gathered matrix against the (replicated) gallery.
    q_local / q_all are torch tensors ([n,512] / [world*n,512] fp16) on the pipeline's stream."""
    pipe.detect(frames_dev, H, W)
    pipe.embed(frames_dev, H, W)            # writes q_local (pipe.q aliases it)
    dist.all_gather_into_tensor(q_all, q_local)
    r = dist.get_rank()
    n = pipe.n_slots
    pipe.match(gallery, thresh, q=q_all[r * n:(r + 1) * n].data_ptr(), n=n)

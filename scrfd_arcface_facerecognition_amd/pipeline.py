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
        self.post = PostProcessor(det.ctx, batch, cap=det_cap)     # (the detector may live on another context / stream than the recogniser: the whole detect stage runs there)
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
                self._det_in = self.det.ctx.empty((self.B, in_h, in_w, 3), np.uint8)
            sc = C.c_double()
            check(self.det.ctx.lib.fid_letterbox(self.det.ctx.handle, _lib._ptr(frames_dev), self.B, H, W,
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
        # empty face slots (f >= counts[b]) become zero rows marked by -0.0 in element 0: they never match, and the matrix the all-gather
        # moves tells every rank which slots of the other ranks hold faces (gathered_face_counts)
        check(self.ctx.lib.fid_l2_normalize_f16_slots(self.ctx.handle, C.c_void_p(emb_ptr), self.n_slots, self.emb_dim,
                                                      C.c_void_p(self.post.counts.ptr), self.F, _lib._ptr(self.q)))

    def match(self, gallery: Gallery, thresh: float, q=None, n=None, idx=None, score=None):
        gallery.match_device(self.q if q is None else q, self.n_slots if n is None else n, thresh,
                             self.idx if idx is None else idx, self.score if score is None else score)

    def match_keys(self, gallery: Gallery, q, n, first_row, keys):
        """this rank's shard of a row-sharded gallery: one packed (score, global index) key per query"""
        check(self.ctx.lib.fid_match_keys(self.ctx.handle, gallery.handle, _lib._ptr(q), int(n), int(first_row), _lib._ptr(keys)))

    def match_merge(self, keys_all, parts, n, gallery_total, thresh, idx, score):
        check(self.ctx.lib.fid_match_merge(self.ctx.handle, _lib._ptr(keys_all), int(parts), int(n), int(gallery_total),
                                           float(thresh), _lib._ptr(idx), _lib._ptr(score)))

    def run_step(self, frames_dev, H, W, gallery: Gallery, thresh: float = 0.4):
        """One full pass over one batch; asynchronous (results stay on the device)."""
        if self.det.ctx is not self.ctx:
            # detect() runs on the detector's stream, embed() on this pipeline's: nothing orders the two.  A staged schedule calls
            # detect / embed / match itself and puts its own events between the streams (bench.py --schedule stages).
            raise ValueError("FacePipeline.run_step: the detector lives on another context / stream; call detect(), embed(), match() "
                             "with your own cross-stream events")
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


class GroupedFacePipeline(FacePipeline):
    """FacePipeline whose recogniser runs once per `group` consecutive steps (dynamic batching across steps, round 5).

    Every step still detects, post-processes and aligns ITS batch of B frames at once (the frames may be overwritten as soon as
    `run_step` has been enqueued); the B*F crops of the group's steps collect in one buffer and the last step of the group sends all
    group*B*F of them through IResNet, the L2 normalisation and ONE gallery match.  IResNet-50 at 64 crops has one tile per CU on its
    26 stage-3 layers (a launch = one latency chain); at 128 crops it runs at 810 instead of 664 TFLOP/s (profiles/r05/steady.txt): the same
    work per face, 18 % less time.  The price is latency: a step's identities exist when its group's last step has run
    (`flush` finishes a partial group).  Slots of step k of the group: [k*B*F, (k+1)*B*F) of `q`, `idx`, `score`."""

    def __init__(self, ctx: Context, det: CompiledNet, rec: CompiledNet, *, batch: int, faces_per_frame: int = 1, group: int = 2,
                 det_cap: int = 256, **kw):
        assert group >= 1 and rec.max_batch >= group * batch * faces_per_frame
        q_buffer = kw.pop("q_buffer", None)
        super().__init__(ctx, det, rec, batch=batch, faces_per_frame=faces_per_frame, det_cap=det_cap, **kw)
        self.group = int(group)
        self.posts = [self.post] + [PostProcessor(det.ctx, batch, cap=det_cap) for _ in range(self.group - 1)]
        n = self.n_slots
        self.crops = ctx.empty((self.group * n, 112, 112, 3), np.uint8)
        self.q = q_buffer if q_buffer is not None else ctx.empty((self.group * n, self.emb_dim), np.float16)
        self.idx = ctx.empty((self.group * n,), np.int32)
        self.score = ctx.empty((self.group * n,), np.float32)
        self.k = 0                                           # steps of the current group already detected + aligned

    def collect(self, frames_dev, H, W) -> bool:
        """detect + post-process + align one batch into the group's crop buffer; True when the group is full"""
        if self.det.ctx is not self.ctx:
            raise ValueError("GroupedFacePipeline: detector and recogniser must share one context / stream")
        k, n = self.k, self.n_slots
        if k >= self.group:
            raise RuntimeError("GroupedFacePipeline.collect: the group is full -- embed_collected() / flush() it first")
        self.post = self.posts[k]
        self.detect(frames_dev, H, W)
        check(self.ctx.lib.fid_align_crops(self.ctx.handle, _lib._ptr(frames_dev), self.B, H, W,
                                           C.c_void_p(self.post.kps.ptr), C.c_void_p(self.post.counts.ptr),
                                           self.post.cap, self.F, C.c_void_p(self.crops.ptr + k * n * 112 * 112 * 3), None))
        self.k += 1
        return self.k == self.group

    def run_step(self, frames_dev, H, W, gallery: Gallery, thresh: float = 0.4):
        if self.collect(frames_dev, H, W):
            self.flush(gallery, thresh)

    def embed_collected(self) -> int:
        """IResNet + L2 normalisation of the steps collected so far into q[0 : steps * B * F]; returns steps (0: nothing collected)"""
        steps, n = self.k, self.n_slots
        if steps == 0:
            return 0
        self.rec.run_device(self.crops, steps * n)
        emb_ptr, _, _ = self.rec.tensor(self.rec.low.outputs[0])
        q_ptr = _lib._ptr(self.q).value
        for j in range(steps):
            check(self.ctx.lib.fid_l2_normalize_f16_slots(self.ctx.handle, C.c_void_p(emb_ptr + j * n * self.emb_dim * 4), n, self.emb_dim,
                                                          C.c_void_p(self.posts[j].counts.ptr), self.F,
                                                          C.c_void_p(q_ptr + j * n * self.emb_dim * 2)))
        self.last_steps, self.k = steps, 0
        return steps

    def flush(self, gallery: Gallery, thresh: float = 0.4):
        """embed + match the steps collected so far (a whole group, or what a stream's end left of one)"""
        steps = self.embed_collected()
        if steps:
            self.match(gallery, thresh, n=steps * self.n_slots)

    def results(self, gallery: Gallery):
        """per step of the last finished group: the list FacePipeline.results returns for one batch"""
        out = []
        idx_all, score_all = self.idx.download(), self.score.download()
        for j in range(getattr(self, "last_steps", 0)):
            post = self.posts[j]
            post.check()
            counts, det, kps = post.counts.download(), post.det.download(), post.kps.download()
            idx = idx_all[j * self.n_slots:(j + 1) * self.n_slots].reshape(self.B, self.F)
            score = score_all[j * self.n_slots:(j + 1) * self.n_slots].reshape(self.B, self.F)
            frames = []
            for b in range(self.B):
                faces = []
                for f in range(min(int(counts[b]), self.F)):
                    i = int(idx[b, f])
                    faces.append((det[b, f, :4].copy(), float(det[b, f, 4]), kps[b, f].reshape(5, 2).copy(),
                                  gallery.names[i] if i >= 0 else "Unknown", float(score[b, f])))
                frames.append(faces)
            out.append(frames)
        return out


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


# ---- gallery construction: reference main.py:78-105 ----------------------------------------------------

def _read_image(path: str) -> Optional[np.ndarray]:
    """cv2.imread stand-in for hosts without OpenCV: returns uint8 BGR [H,W,3] or None (like imread on a file
    it cannot decode).  Uses OpenCV when importable; otherwise understands .npy arrays and binary PPM (P6)."""
    try:
        import cv2
        return cv2.imread(path)
    except ImportError:
        pass
    try:
        if path.endswith(".npy"):
            a = np.load(path, allow_pickle=False)
            return np.ascontiguousarray(a, dtype=np.uint8) if a.ndim == 3 and a.shape[2] == 3 else None
        with open(path, "rb") as f:
            data = f.read()
        if data[:2] == b"P6":
            tok, pos = [], 2
            while len(tok) < 3:                          # width, height, maxval (comments allowed)
                while data[pos:pos + 1].isspace():
                    pos += 1
                if data[pos:pos + 1] == b"#":
                    pos = data.index(b"\n", pos) + 1
                    continue
                end = pos
                while not data[end:end + 1].isspace():
                    end += 1
                tok.append(int(data[pos:end])); pos = end
            w, h, mx = tok
            if mx != 255:
                return None
            rgb = np.frombuffer(data, np.uint8, count=w * h * 3, offset=pos + 1).reshape(h, w, 3)
            return np.ascontiguousarray(rgb[..., ::-1])
    except Exception:
        return None
    return None


def build_targets_from_images(detector, recognizer, images: Sequence[np.ndarray], names: Sequence[str],
                              paths: Optional[Sequence[str]] = None) -> List[Tuple[np.ndarray, str]]:
    """The body of reference build_targets (main.py:91-103) for images already in memory:
    per image `detect(image, max_num=1)` (:96), images without a face are skipped with a warning (:98-100),
    the best face is embedded (`recognizer(image, kpss[0])`, :102) and `(embedding, name)` collected (:103),
    in input order.  Images of one shape go through the detector, the alignment and the recogniser as ONE
    batch each (detect_batch -> fid_align_crops -> get_feat); objects without the batched surface are driven
    through the reference's per-image calls."""
    import logging
    assert len(images) == len(names)
    kps_of: List[Optional[np.ndarray]] = [None] * len(images)
    batched = hasattr(detector, "detect_batch") and hasattr(recognizer, "get_feat") and getattr(recognizer, "_native", False)
    by_shape = {}
    for i, im in enumerate(images):
        by_shape.setdefault(tuple(im.shape), []).append(i)
    emb_of: List[Optional[np.ndarray]] = [None] * len(images)
    for shape, ids in by_shape.items():
        if not batched:
            for i in ids:
                _, kpss = detector.detect(images[i], max_num=1)
                if len(kpss):
                    emb_of[i] = np.asarray(recognizer(images[i], kpss[0]), dtype=np.float32).reshape(-1)
            continue
        stack = np.ascontiguousarray(np.stack([images[i] for i in ids]), dtype=np.uint8)
        dets = detector.detect_batch(stack, max_num=1)
        hit = [k for k, (_, kpss) in enumerate(dets) if len(kpss)]
        if not hit:
            continue
        ctx = recognizer.ctx
        H, W = shape[:2]
        n = len(hit)
        kps = np.stack([dets[k][1][0].reshape(10) for k in hit]).astype(np.float32).reshape(n, 1, 10)
        fr = ctx.to_device(stack[hit])
        kp, cn = ctx.to_device(kps), ctx.to_device(np.ones(n, np.int32))
        crops = ctx.empty((n, 112, 112, 3), np.uint8)
        check(ctx.lib.fid_align_crops(ctx.handle, C.c_void_p(fr.ptr), n, H, W, C.c_void_p(kp.ptr), C.c_void_p(cn.ptr),
                                      1, 1, C.c_void_p(crops.ptr), None))
        feats = recognizer.get_feat(crops.download())
        for k, e in zip(hit, feats):
            emb_of[ids[k]] = np.ascontiguousarray(e, dtype=np.float32).reshape(-1)
    targets = []
    for i, name in enumerate(names):
        if emb_of[i] is None:
            logging.warning(f"No face detected in {paths[i] if paths else name}. Skipping...")
            continue
        targets.append((emb_of[i], name))
    return targets


def build_targets(detector, recognizer, params, *, loader=None) -> List[Tuple[np.ndarray, str]]:
    """reference main.py:78-105, same signature: `params.faces_dir` is scanned with os.listdir (:91),
    `name = filename[:-4]` (:92), every file is read (:95) and handed to build_targets_from_images.
    `params` may also be the directory path itself.  Files the loader cannot decode are skipped."""
    import logging
    import os
    faces_dir = params if isinstance(params, str) else params.faces_dir
    loader = loader or _read_image
    images, names, paths = [], [], []
    for filename in os.listdir(faces_dir):
        image_path = os.path.join(faces_dir, filename)
        image = loader(image_path)
        if image is None:
            logging.warning(f"Cannot read {image_path}. Skipping...")
            continue
        images.append(image); names.append(filename[:-4]); paths.append(image_path)
    return build_targets_from_images(detector, recognizer, images, names, paths)


def gallery_from_targets(ctx: Context, targets: Sequence[Tuple[np.ndarray, str]]) -> Gallery:
    """The `targets` list as a device-resident Gallery (what frame_processor scans, main.py:136-142)."""
    assert len(targets) > 0, "no targets: every gallery image was skipped"
    return Gallery(ctx, np.stack([t[0] for t in targets]).astype(np.float32), [t[1] for t in targets])


# ---- multi-GPU step: the single collective -----------------------------------------------------

class Communicator:
    """fid_comm: this rank's end of an RCCL communicator owned by libfaceid (include/faceid.h), so the collective runs
    behind the C-ABI on the context's stream.  `exchange_id(id_bytes_or_None) -> id_bytes` is how the host hands rank 0's
    128-byte rendezvous id to every rank (a torch.distributed broadcast, an MPI bcast, a file ...)."""

    def __init__(self, ctx: Context, world: int, rank: int, exchange_id):
        self.ctx, self.world, self.rank = ctx, int(world), int(rank)
        n = 128
        buf = (C.c_uint8 * n)()
        if rank == 0:
            check(ctx.lib.fid_comm_unique_id(buf, n))
        ident = exchange_id(bytes(buf) if rank == 0 else None)
        assert len(ident) == n
        h = C.c_void_p()
        check(ctx.lib.fid_comm_init_rank(ctx.handle, self.world, self.rank, ident, n, C.byref(h)))
        self.handle = h

    def get_rank(self):
        return self.rank

    def get_world_size(self):
        return self.world

    def all_gather_into_tensor(self, out, inp):
        """same call shape as torch.distributed's; `out` / `inp` are device buffers (anything _lib._ptr accepts)
        and `inp` carries .nbytes or (numel, element_size)"""
        nbytes = inp.nbytes if hasattr(inp, "nbytes") and not callable(inp.nbytes) else inp.numel() * inp.element_size()
        check(self.ctx.lib.fid_allgather(self.ctx.handle, self.handle, _lib._ptr(inp), _lib._ptr(out), int(nbytes)))

    def close(self):
        if self.handle:
            self.ctx.lib.fid_comm_destroy(self.ctx.handle, self.handle)
            self.handle = None


EMPTY_SLOT_MARK = 0x8000          # fp16 -0.0 in element 0 of an otherwise all +0.0 row (csrc/match.hip l2norm_rows)


def empty_slot_rows(q_all: np.ndarray) -> np.ndarray:
    """bool per row of a unit-embedding matrix (fp16, host copy): True = the row is the EMPTY-slot marker fid_l2_normalize_f16_slots writes
    for face slots f >= counts[b] (-0.0 then +0.0s).  A degenerate face inside a frame's valid prefix (zero / NaN / inf embedding) is an
    all +0.0 row instead: numerically the same zero row (never matches), but still a face."""
    u = np.ascontiguousarray(np.asarray(q_all)).view(np.uint16).reshape(-1, np.asarray(q_all).shape[-1])
    return (u[:, 0] == EMPTY_SLOT_MARK) & ~u[:, 1:].any(axis=1)


def gathered_face_counts(q_all: np.ndarray, frames_total: int, faces_per_frame: int) -> np.ndarray:
    """Face count of every frame of the WHOLE batch from the gathered unit-embedding matrix [frames_total * F, 512] (fp16 host copy):
    a slot holds a face iff its row is not the empty-slot marker (empty_slot_rows) -- so the one collective of SURVEY.md 8e also carries
    `counts[]` EXACTLY and idx_all / score_all can be read like the reference's per-face loop (main.py:132-134), which visits a detected
    face whether or not its embedding is degenerate.  (Round 3 counted non-zero rows: a degenerate face at slot f < count shifted every
    later slot of its frame by one.)"""
    q = np.asarray(q_all)
    assert q.dtype == np.float16, "the gathered matrix is fp16 (bit patterns matter: -0.0 marks an empty slot)"
    return (~empty_slot_rows(q)).reshape(frames_total, faces_per_frame).sum(axis=1).astype(np.int32)


def _slice_ptr(buf, row0: int, row_bytes: int):
    """device address of row `row0` of a 2-D device buffer (torch tensor / DeviceBuffer / raw int)"""
    base = buf.data_ptr() if hasattr(buf, "data_ptr") else (buf.ptr if hasattr(buf, "ptr") else int(buf))
    return base + row0 * row_bytes


def _first_rows(buf, rows: int):
    """the first `rows` rows of a device buffer the collective / match calls accept (torch tensor or numpy array: a view; else unchanged)"""
    if buf is None or not hasattr(buf, "shape") or int(buf.shape[0]) == rows:
        return buf
    return buf[:rows]


def run_step_distributed(pipe, frames_dev, H, W, gallery, thresh, q_local, q_all, dist, *, idx_all=None, score_all=None,
                         match_scope: str = "all", keys_local=None, keys_all=None, gallery_first_row: int = 0,
                         gallery_total: int = 0, flush: bool = False):
    """One multi-GPU step: local detect / align / embed on this rank's frames, ONE all-gather of the
    unit fp16 embeddings (the collective BASELINE.json's north_star names), then the gallery match on the gathered
    matrix.  `dist` is torch.distributed (RCCL as backend "nccl", gloo in the CPU tests) or a `Communicator`
    (the C-ABI's own RCCL communicator): anything with get_rank / get_world_size / all_gather_into_tensor.

    q_local [n,512] / q_all [world*n,512] fp16 device buffers (pipe.q aliases q_local).

    match_scope
      "all"      (default; replicated gallery) every rank matches ALL world*n gathered queries, so every rank holds the
                 whole batch's result list in idx_all / score_all [world*n] with no second collective -- the gather's
                 output is what the match reads.
      "own"      every rank matches only its own n rows of the gathered matrix into pipe.idx / pipe.score (the cheapest
                 form when the host collects per-rank results itself).
      "sharded"  gallery sharded by contiguous row blocks (the 1 M-entry variant): `gallery` holds rows
                 [gallery_first_row, +G_local) of a gallery_total-row gallery; every rank scans its shard for all
                 world*n queries (fid_match_keys), a second tiny all-gather exchanges the packed (score, index) keys
                 (keys_local [world*n] u64 -> keys_all [world, world*n]), fid_match_merge takes the arg-max.

    A GroupedFacePipeline (anything with collect / embed_collected) only detects + aligns until its group is full: the call returns False
    and nothing is exchanged.  The group's last step (or `flush=True`, which embeds what a stream's end left of a group without detecting
    anything) runs IResNet on all collected crops, then the SAME one all-gather and match on n = steps * B * F rows per rank -- the buffers
    are sized for a full group, a partial group uses their first rows (every rank holds the same number of steps).  Returns True when results
    were written.
    """
    r, world = dist.get_rank(), dist.get_world_size()
    n = pipe.n_slots
    grouped = hasattr(pipe, "collect")
    n_cap = n * (pipe.group if grouped else 1)              # rows per rank the buffers must hold

    def rows(buf):                                   # entries of a result / key buffer (torch tensor, DeviceBuffer, numpy array)
        if buf is None:
            return 0
        return int(np.prod(tuple(buf.shape))) if hasattr(buf, "shape") else int(buf.nbytes // buf.itemsize)

    # the scopes that write world*n results must be given buffers of that size: falling back to pipe.idx / pipe.score (n entries)
    # would overrun them at world > 1 (a one-rank group may use them: world*n == n)
    if match_scope == "all" and idx_all is None and score_all is None and world == 1:
        idx_all, score_all = pipe.idx, pipe.score
    if match_scope in ("all", "sharded") and (rows(idx_all) < world * n_cap or rows(score_all) < world * n_cap):
        raise ValueError(f"match_scope={match_scope!r} writes {world * n_cap} results: pass idx_all / score_all with at least that many entries")
    if match_scope == "sharded" and (rows(keys_local) < world * n_cap or rows(keys_all) < world * world * n_cap or gallery_total <= 0):
        raise ValueError("match_scope='sharded' needs keys_local [world*n], keys_all [world*world*n] (uint64) and gallery_total")
    if grouped:
        if not flush and not pipe.collect(frames_dev, H, W):
            return False                    # the group is still open: this step's crops wait for the group's last step
        steps = pipe.embed_collected()      # writes q_local[0 : steps * B * F]
        if steps == 0:
            return False
        n = steps * pipe.n_slots
        q_local, q_all = _first_rows(q_local, n), _first_rows(q_all, world * n)
        idx_all, score_all = _first_rows(idx_all, world * n), _first_rows(score_all, world * n)
        keys_local, keys_all = _first_rows(keys_local, world * n), _first_rows(keys_all, world * world * n)
    else:
        pipe.detect(frames_dev, H, W)
        pipe.embed(frames_dev, H, W)        # writes q_local (pipe.q aliases it); empty face slots are zero rows
    dist.all_gather_into_tensor(q_all, q_local)
    if match_scope == "own":
        pipe.match(gallery, thresh, q=_slice_ptr(q_all, r * n, 512 * 2), n=n)
    elif match_scope == "all":
        pipe.match(gallery, thresh, q=q_all, n=world * n, idx=idx_all, score=score_all)
    elif match_scope == "sharded":
        pipe.match_keys(gallery, q_all, world * n, gallery_first_row, keys_local)
        dist.all_gather_into_tensor(keys_all, keys_local)
        pipe.match_merge(keys_all, world, world * n, gallery_total, thresh, idx_all, score_all)
    else:
        raise ValueError(f"unknown match_scope {match_scope!r}")
    return True

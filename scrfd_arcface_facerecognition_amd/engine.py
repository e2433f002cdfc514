"""Host-side handles over the C-ABI: a compiled conv net, the SCRFD post-process, alignment and the
gallery.  Everything here only moves pointers and shapes; all arithmetic happens in libfaceid.so."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import Context, DeviceBuffer, check
from .archs import ARCHS, ONNX_BASENAMES, Net, synth_params
from .lower import Lowered, lower


class CompiledNet:
    """fid_net: layer table + packed weights resident on one device.  The session.run replacement
    (reference models/scrfd.py:83, models/arcface.py:51)."""

    def __init__(self, ctx: Context, net: Net, params: Dict[str, np.ndarray], max_batch: int = 64):
        self.ctx = ctx
        self.net = net
        self.low: Lowered = lower(net, params)
        self.max_batch = int(max_batch)
        ops = np.ascontiguousarray(self.low.ops, dtype=np.int32)
        tens = np.ascontiguousarray(self.low.tensors, dtype=np.int32)
        h = C.c_void_p()
        blob = self.low.blob
        check(ctx.lib.fid_net_create(ctx.handle, ops.ctypes.data_as(_lib.c_i32_p), ops.shape[0],
                                     tens.ctypes.data_as(_lib.c_i32_p), tens.shape[0], blob, len(blob),
                                     net.in_hw[0], net.in_hw[1], self.max_batch, C.byref(h)))
        self.handle = h
        self.in_hw = tuple(net.in_hw)
        self._in_buf: Optional[DeviceBuffer] = None

    # -- running ----------------------------------------------------------------------------
    def run_device(self, images_dev, batch: int):
        """images_dev: device pointer to uint8 BGR [batch, H, W, 3]"""
        check(self.ctx.lib.fid_net_run(self.ctx.handle, self.handle, _lib._ptr(images_dev), int(batch)))

    def run(self, images: np.ndarray):
        """images: uint8 [B,H,W,3] host array (B <= max_batch)."""
        images = np.ascontiguousarray(images, dtype=np.uint8)
        assert images.ndim == 4 and images.shape[1:3] == self.in_hw and images.shape[3] == 3, images.shape
        if self._in_buf is None or self._in_buf.nbytes < images.nbytes:
            self._in_buf = self.ctx.empty((self.max_batch,) + images.shape[1:], np.uint8)
        check(self.ctx.lib.fid_memcpy_h2d(self.ctx.handle, C.c_void_p(self._in_buf.ptr),
                                          images.ctypes.data_as(C.c_void_p), images.nbytes))
        self.ctx.sync()
        self.run_device(self._in_buf, images.shape[0])
        return images.shape[0]

    def run_profiled(self, images_dev, batch: int) -> np.ndarray:
        ms = np.zeros(len(self.low.op_names), dtype=np.float32)
        check(self.ctx.lib.fid_net_run_profiled(self.ctx.handle, self.handle, _lib._ptr(images_dev), int(batch),
                                                ms.ctypes.data_as(_lib.c_f32_p)))
        return ms

    # -- tensors ----------------------------------------------------------------------------
    def tensor(self, name: str):
        """(device pointer, (H, W, C, C_stored), dtype) of a tensor of the last run."""
        p = C.c_void_p()
        dims = (C.c_int * 4)()
        dt = C.c_int()
        check(self.ctx.lib.fid_net_tensor(self.handle, self.low.tensor_id[name], C.byref(p), dims, C.byref(dt)))
        return p.value, tuple(dims), (np.float32 if dt.value == 1 else np.float16)

    def read(self, name: str, batch: int) -> np.ndarray:
        """Download tensor `name` as float32 [batch, H, W, C] (channel padding stripped)."""
        ptr, (H, W, Cc, Cp), dt = self.tensor(name)
        buf = self.ctx.borrow(ptr, (batch, H, W, Cp), dt)
        return buf.download()[..., :Cc].astype(np.float32)

    def save_plan(self, path: str):
        """append this net's kernel picks (per conv op and batch size) to a plan file"""
        check(self.ctx.lib.fid_net_plan_save(self.handle, str(path).encode()))

    def load_plan(self, path: str) -> int:
        """install the picks of a plan file that match this device and layer table; returns how many"""
        n = C.c_int()
        check(self.ctx.lib.fid_net_plan_load(self.handle, str(path).encode(), C.byref(n)))
        return n.value

    def plans(self) -> List[dict]:
        """the kernel pick of every conv op that has run so far: [{op, name, batch, gen, bm, bn, bk, ksplit, ns}] (parsed from
        fid_net_plan_save's lines; tests use it to check WHICH kernel family produced a result)"""
        import os
        import tempfile
        fd, path = tempfile.mkstemp(suffix=".plan")
        os.close(fd)
        try:
            self.save_plan(path)
            out = []
            with open(path) as f:
                for line in f:
                    parts = line.rstrip("\n").split("|")
                    if len(parts) != 5:
                        continue
                    v = [int(x) for x in parts[4].split()]
                    oi = int(parts[2])
                    out.append(dict(op=oi, name=self.low.op_names[oi], batch=int(parts[3]), gen=v[0], bm=v[1], bn=v[2], bk=v[3],
                                    ksplit=v[4], ns=v[5]))
            return out
        finally:
            os.unlink(path)

    def macs_per_image(self) -> float:
        v = C.c_double()
        check(self.ctx.lib.fid_net_macs(self.handle, C.byref(v)))
        return v.value

    def close(self):
        if self.handle:
            self.ctx.lib.fid_net_destroy(self.ctx.handle, self.handle)
            self.handle = None


def resolve_model(model_path: str, in_hw=None):
    """model_path -> (Net, params).  Accepted:
      'synthetic:<arch>[?seed=N]'   seeded random-init weights of a known architecture
      '<file>.npz'                  parameters saved by save_params() (+ key '__arch__')
      '<file>.onnx'                 a real ONNX file (weights read by onnx_reader, no onnx/ORT needed)
    Like onnxruntime, a missing file raises (reference models/scrfd.py:66-68 prints and re-raises)."""
    import os
    if model_path is None:
        raise ValueError("model_path is required")
    if model_path.startswith("synthetic:"):
        spec = model_path[len("synthetic:"):]
        arch, _, q = spec.partition("?")
        seed = 0
        for kv in q.split("&"):
            if kv.startswith("seed="):
                seed = int(kv[5:])
        if arch not in ARCHS:
            raise ValueError(f"unknown architecture {arch!r}; known: {sorted(ARCHS)}")
        net = ARCHS[arch](in_hw) if in_hw else ARCHS[arch]()
        return net, synth_params(net, seed)
    if not os.path.exists(model_path):
        raise FileNotFoundError(f"model file not found: {model_path}")
    if model_path.endswith(".npz"):
        z = np.load(model_path, allow_pickle=False)
        arch = str(z["__arch__"])
        net = ARCHS[arch](in_hw) if in_hw else ARCHS[arch]()
        return net, {k: z[k] for k in z.files if k != "__arch__"}
    if model_path.endswith(".onnx"):
        from .onnx_reader import load_onnx_model
        return load_onnx_model(model_path, in_hw)
    raise ValueError(f"unsupported model file {model_path}")


def save_params(path: str, arch: str, params: Dict[str, np.ndarray]):
    np.savez(path, __arch__=np.array(arch), **params)


class HeadViews:
    """The 9 strided views fid_scrfd_postprocess reads (include/faceid.h)."""

    def __init__(self, ptrs: Sequence[int], pix: Sequence[int], anc: Sequence[int], bstride: Sequence[int]):
        self.ptrs = (C.c_void_p * 9)(*[C.c_void_p(int(p)) for p in ptrs])
        self.pix = (C.c_int32 * 9)(*[int(v) for v in pix])
        self.anc = (C.c_int32 * 9)(*[int(v) for v in anc])
        self.bstride = (C.c_int64 * 9)(*[int(v) for v in bstride])

    @staticmethod
    def from_onnx_layout(bufs: Sequence[DeviceBuffer], A: int = 2):
        """bufs: 9 device arrays [B, N_l, 1|4|10] in the ONNX output order (scores, bbox, kps)."""
        ptrs, pix, anc, bs = [], [], [], []
        for k, b in enumerate(bufs):
            c = (1, 4, 10)[k // 3]
            ptrs.append(b.ptr)
            pix.append(A * c)
            anc.append(c)
            bs.append(b.shape[1] * c)
        return HeadViews(ptrs, pix, anc, bs)

    @staticmethod
    def from_fused(cnet: CompiledNet):
        """Views into the executor's fused fp32 head tensors [B, H, W, 32]."""
        ptrs, pix, anc, bs = [None] * 9, [0] * 9, [0] * 9, [0] * 9
        for li, name in enumerate(cnet.low.outputs):
            h = cnet.low.heads[name]
            base, (H, W, _, Cp), dt = cnet.tensor(name)
            assert dt == np.float32
            for part, (off, c) in enumerate((h["score"], h["bbox"], h["kps"])):
                k = part * 3 + li
                ptrs[k] = base + off * 4
                pix[k] = Cp
                anc[k] = c
                bs[k] = H * W * Cp
        return HeadViews(ptrs, pix, anc, bs)


class PostProcessor:
    """fid_scrfd_postprocess with persistent output buffers."""

    def __init__(self, ctx: Context, max_batch: int, cap: int = 256, cand_cap: int = 4096):
        self.ctx, self.cap, self.max_batch = ctx, int(cap), int(max_batch)
        self.det = ctx.empty((max_batch, cap, 5), np.float32)
        self.kps = ctx.empty((max_batch, cap, 10), np.float32)
        self.counts = ctx.empty((max_batch,), np.int32)
        self.cand_cap = int(cand_cap)
        check(ctx.lib.fid_scrfd_set_candidate_capacity(ctx.handle, self.cand_cap))

    def run(self, hv: HeadViews, B, in_hw, img_hw, conf, iou, max_num=0, metric=0, A=2):
        assert B <= self.max_batch
        check(self.ctx.lib.fid_scrfd_postprocess(
            self.ctx.handle, C.cast(hv.ptrs, _lib.c_void_pp), hv.pix, hv.anc, hv.bstride, int(B), int(in_hw[0]),
            int(in_hw[1]), int(A), int(img_hw[0]), int(img_hw[1]), float(conf), float(iou), int(max_num),
            int(metric), C.c_void_p(self.det.ptr), C.c_void_p(self.kps.ptr), C.c_void_p(self.counts.ptr), self.cap))

    def check(self) -> int:
        m = C.c_int()
        check(self.ctx.lib.fid_scrfd_check(self.ctx.handle, C.byref(m)))
        return m.value

    def fetch(self, B) -> List:
        """[(det[K,5], kps[K,5,2])] per frame, host arrays (synchronises)."""
        self.check()
        counts = self.counts.download()[:B]
        det = self.det.download()
        kps = self.kps.download()
        return [(det[b, :counts[b]].copy(), kps[b, :counts[b]].reshape(-1, 5, 2).copy()) for b in range(B)]


class Gallery:
    """fid_gallery: unit-length fp16 rows in HBM (what build_targets collects, reference main.py:78-105)."""

    def __init__(self, ctx: Context, embeddings: np.ndarray, names: Optional[Sequence[str]] = None):
        emb = np.ascontiguousarray(embeddings, dtype=np.float32)
        assert emb.ndim == 2
        self.ctx = ctx
        self.G, self.dim = emb.shape
        self.names = list(names) if names is not None else [str(i) for i in range(self.G)]
        h = C.c_void_p()
        check(ctx.lib.fid_gallery_create(ctx.handle, emb.ctypes.data_as(C.c_void_p), self.G, self.dim, C.byref(h)))
        self.handle = h
        gp = C.c_int()
        check(ctx.lib.fid_gallery_info(h, None, C.byref(gp), None))
        self.Gp = gp.value

    def match_device(self, q_f16_dev, n, thresh, idx_dev, score_dev):
        check(self.ctx.lib.fid_match(self.ctx.handle, self.handle, _lib._ptr(q_f16_dev), int(n), float(thresh),
                                     _lib._ptr(idx_dev), _lib._ptr(score_dev)))

    def close(self):
        if self.handle:
            self.ctx.lib.fid_gallery_destroy(self.ctx.handle, self.handle)
            self.handle = None


class VectorGallery:
    """An updatable gallery with top-k search: the role QdrantManager plays in the reference's product layer
    (qdrant_manager.py:91-212: add_embedding / search_similar(limit, score_threshold) / delete_embedding),
    kept in HBM as unit fp16 rows.  Ids are arbitrary hashables; rows freed by delete() are reused."""

    def __init__(self, ctx: Context, dim: int = 512, capacity: int = 1024):
        self.ctx, self.dim = ctx, int(dim)
        self._gal = Gallery(ctx, np.zeros((int(capacity), dim), np.float32))
        self.row_of: Dict[object, int] = {}
        self.id_of: Dict[int, object] = {}
        self._free = list(range(int(capacity) - 1, -1, -1))

    def __len__(self):
        return len(self.row_of)

    def _grow(self):
        old = self._gal
        cap = old.G * 2
        rows = self.ctx.borrow(_gallery_ptr(old), (old.Gp, self.dim), np.float16).download()[:old.G].astype(np.float32)
        new = Gallery(self.ctx, np.concatenate([rows, np.zeros((cap - old.G, self.dim), np.float32)]))
        self._free = list(range(cap - 1, old.G - 1, -1)) + self._free
        old.close()
        self._gal = new

    def upsert(self, ids, embeddings):
        emb = np.ascontiguousarray(embeddings, dtype=np.float32).reshape(len(ids), self.dim)
        rows = []
        for i in ids:
            if i not in self.row_of:
                if not self._free:
                    self._grow()
                r = self._free.pop()
                self.row_of[i], self.id_of[r] = r, i
            rows.append(self.row_of[i])
        rows = np.asarray(rows, dtype=np.int32)
        check(self.ctx.lib.fid_gallery_set_rows(self.ctx.handle, self._gal.handle, rows.ctypes.data_as(_lib.c_i32_p),
                                                emb.ctypes.data_as(C.c_void_p), len(rows)))

    def delete(self, ids):
        rows = np.asarray([self.row_of.pop(i) for i in ids], dtype=np.int32)
        for r in rows:
            self.id_of.pop(int(r))
            self._free.append(int(r))
        zeros = np.zeros((len(rows), self.dim), np.float32)
        check(self.ctx.lib.fid_gallery_set_rows(self.ctx.handle, self._gal.handle, rows.ctypes.data_as(_lib.c_i32_p),
                                                zeros.ctypes.data_as(C.c_void_p), len(rows)))

    def search(self, embeddings, k: int = 5, score_threshold: float = 0.0):
        """-> per query a list of (id, score), best first (at most k, only scores > max(0, threshold))."""
        emb = np.ascontiguousarray(embeddings, dtype=np.float32).reshape(-1, self.dim)
        n = emb.shape[0]
        e = self.ctx.to_device(emb)
        q = self.ctx.empty((n, self.dim), np.float16)
        check(self.ctx.lib.fid_l2_normalize_f16(self.ctx.handle, C.c_void_p(e.ptr), n, self.dim, C.c_void_p(q.ptr)))
        idx, sc = self.ctx.empty((n, k), np.int32), self.ctx.empty((n, k), np.float32)
        check(self.ctx.lib.fid_gallery_topk(self.ctx.handle, self._gal.handle, C.c_void_p(q.ptr), n, int(k), float(score_threshold),
                                            C.c_void_p(idx.ptr), C.c_void_p(sc.ptr)))
        I, S = idx.download(), sc.download()
        return [[(self.id_of[int(j)], float(s)) for j, s in zip(I[r], S[r]) if j >= 0] for r in range(n)]


    # ---- the product layer's duplicate logic on top of search ("duplicate check @0.95, merge @0.8") ---------------------------------
    def is_duplicate(self, embedding, duplicate_threshold: float = 0.95) -> bool:
        """the vector half of the reference's `is_duplicate_image` (smart_face_recognition.py:2632-2641; config.json
        `duplicate_similarity_threshold`): does the nearest stored embedding reach the threshold?"""
        return len(self) > 0 and len(self.search(np.asarray(embedding, np.float32).reshape(1, self.dim), k=1, score_threshold=duplicate_threshold)[0]) > 0

    def similarity_rows(self, ids) -> np.ndarray:
        """cosine similarity of the stored embeddings `ids` against EVERY row of the store, fp32 [len(ids), capacity] (free rows: 0), computed on the
        device in one GEMM per 1 024 ids (fid_cosine_matrix with the store's own unit rows as queries)"""
        gal = self._gal
        base = _gallery_ptr(gal)
        out = np.empty((len(ids), gal.G), np.float32)
        rows = np.asarray([self.row_of[i] for i in ids], dtype=np.int64)
        stage = self.ctx.empty((min(1024, max(1, len(ids))), self.dim), np.float16)
        cm = self.ctx.empty((stage.shape[0], gal.Gp), np.float32)
        unit = self.ctx.borrow(base, (gal.Gp, self.dim), np.float16).download()          # (host copy of the unit rows: the gather of arbitrary ids)
        for c0 in range(0, len(ids), stage.shape[0]):
            blk = rows[c0:c0 + stage.shape[0]]
            q = np.zeros(stage.shape, np.float16)
            q[:len(blk)] = unit[blk]
            stage.upload(q)
            check(self.ctx.lib.fid_cosine_matrix(self.ctx.handle, gal.handle, C.c_void_p(stage.ptr), len(blk), C.c_void_p(cm.ptr)))
            out[c0:c0 + len(blk)] = cm.download()[:len(blk), :gal.G]
        return out

    def find_and_merge_duplicates(self, similarity_threshold: float = 0.8):
        """The reference's `find_and_merge_duplicates` (smart_face_recognition.py:2726-2797; config.json `merge_duplicate_threshold`) on the vector
        store: ids in ascending order; every id that is still stored absorbs all LARGER ids whose similarity reaches the threshold (their rows are
        deleted, as merge_duplicate_persons does through delete_embedding).  The G x G similarities come from the device in one pass (deletions only
        remove candidates, no embedding changes); the greedy pass over them is the reference's loop.  Returns [(kept id, deleted id, similarity)]."""
        ids = sorted(self.row_of)
        if len(ids) < 2:
            return []
        sims = self.similarity_rows(ids)
        col = np.asarray([self.row_of[i] for i in ids])
        alive = {i: True for i in ids}
        merges = []
        for a, p1 in enumerate(ids):
            if not alive[p1]:
                continue
            s = sims[a, col]
            for b in np.argsort(-s, kind="stable"):
                if s[b] < similarity_threshold:
                    break
                p2 = ids[int(b)]
                if p2 <= p1 or not alive[p2]:
                    continue
                alive[p2] = False
                merges.append((p1, p2, float(s[b])))
        if merges:
            self.delete([m[1] for m in merges])
        return merges


def _gallery_ptr(gal: Gallery) -> int:
    p = C.c_void_p()
    check(gal.ctx.lib.fid_gallery_data(gal.handle, C.byref(p)))
    return int(p.value)

"""Batched video front end: the step BEFORE the hot path.  The reference reads one frame,
processes it, reads the next (main.py:174-184; main2.py:91-99 for two cameras).  Here frames are collected
into batches in pinned host memory and uploaded on a separate HIP stream while the previous batch is being
processed (double buffering); non-640x640 frames (e.g. 1080p) are letterboxed on the device by the pipeline."""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Iterator, List

import numpy as np

from ._lib import Context, check
from .engine import Gallery
from .pipeline import FacePipeline


class _Pinned:
    def __init__(self, ctx: Context, shape, dtype):
        self.ctx = ctx
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        check(ctx.lib.fid_pinned_alloc(ctx.handle, self.nbytes, C.byref(p)))
        self.ptr = p.value
        self.array = np.ctypeslib.as_array((C.c_uint8 * self.nbytes).from_address(self.ptr)).view(dtype).reshape(shape)

    def free(self):
        if self.ptr:
            self.ctx.lib.fid_pinned_free(self.ctx.handle, C.c_void_p(self.ptr))
            self.ptr = None


class StreamRunner:
    """Runs `pipe` over an iterable of uint8 BGR frames of one size, `pipe.B` frames per step, yielding
    per-frame face lists (FacePipeline.results) in frame order.  Upload of batch i+1 overlaps compute of batch i."""

    def __init__(self, pipe: FacePipeline, gallery: Gallery, frame_hw, similarity_thresh: float = 0.4):
        self.pipe, self.gallery, self.thr = pipe, gallery, float(similarity_thresh)
        self.H, self.W = int(frame_hw[0]), int(frame_hw[1])
        ctx = pipe.ctx
        shape = (pipe.B, self.H, self.W, 3)
        self.host = [_Pinned(ctx, shape, np.uint8) for _ in range(2)]
        self.dev = [ctx.empty(shape, np.uint8) for _ in range(2)]

    def _upload(self, k: int):
        """copy pinned buffer k -> device buffer k on the upload stream; waits only for the last step that read dev[k]"""
        ctx = self.pipe.ctx
        check(ctx.lib.fid_upload_async_slot(ctx.handle, k, C.c_void_p(self.dev[k].ptr), C.c_void_p(self.host[k].ptr),
                                            self.host[k].nbytes))

    def run(self, frames: Iterable[np.ndarray]) -> Iterator[List]:
        ctx, B = self.pipe.ctx, self.pipe.B
        it = iter(frames)

        def fill(k):
            n = 0
            for f in it:
                self.host[k].array[n] = f
                n += 1
                if n == B:
                    break
            if 0 < n < B:
                self.host[k].array[n:] = 0          # ragged tail: black frames, their results are dropped
            return n

        cur = 0
        n_cur = fill(cur)
        if n_cur:
            self._upload(cur)
        while n_cur:
            check(ctx.lib.fid_upload_wait_slot(ctx.handle, cur))   # compute waits for batch `cur` (no host sync)
            self.pipe.run_step(self.dev[cur], self.H, self.W, self.gallery, self.thr)
            check(ctx.lib.fid_upload_release(ctx.handle, cur))     # dev[cur] is free again once this step has run
            nxt = cur ^ 1
            n_nxt = fill(nxt)                                       # host fills the other pinned buffer meanwhile
            if n_nxt:
                self._upload(nxt)                                   # dev[nxt]'s last reader was step i-1: starts NOW, beside step i
            res = self.pipe.results(self.gallery)                   # synchronises on batch `cur`
            for r in res[:n_cur]:
                yield r
            cur, n_cur = nxt, n_nxt

    def close(self):
        for h in self.host:
            h.free()

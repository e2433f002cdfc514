"""Geometry / similarity helpers with the names and signatures of reference utils/helpers.py
(:6-123).  Each one is a thin host wrapper: upload, one libfaceid call, download.  There is no
numpy implementation behind them -- without the HIP library they raise."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .._lib import Context, check, default_context

reference_alignment = np.array(
    [[[38.2946, 51.6963], [73.5318, 51.5014], [56.0252, 71.7366], [41.5493, 92.3655], [70.7299, 92.2041]]],
    dtype=np.float32)


def _ctx(ctx) -> Context:
    return ctx or default_context(0)


def _align(image, landmark, ctx, want_crop):
    landmark = np.asarray(landmark, dtype=np.float32)
    assert landmark.shape == (5, 2)
    ctx = _ctx(ctx)
    image = np.ascontiguousarray(image, dtype=np.uint8)
    H, W = image.shape[:2]
    fr = ctx.to_device(image[None])
    kp = ctx.to_device(landmark.reshape(1, 1, 10))
    cn = ctx.to_device(np.array([1], np.int32))
    crop = ctx.empty((1, 112, 112, 3), np.uint8)
    M = ctx.empty((1, 6), np.float64)
    check(ctx.lib.fid_align_crops(ctx.handle, C.c_void_p(fr.ptr), 1, H, W, C.c_void_p(kp.ptr), C.c_void_p(cn.ptr), 1, 1,
                                  C.c_void_p(crop.ptr), C.c_void_p(M.ptr)))
    return (crop.download()[0] if want_crop else None), M.download().reshape(2, 3)


def estimate_norm(landmark, image_size=112, *, ctx=None):
    """helpers.py:18-53 -> (M float64 [2,3], 0)."""
    if image_size != 112:
        raise NotImplementedError("only the 112x112 ArcFace template is implemented on the device")
    _, M = _align(np.zeros((2, 2, 3), np.uint8), landmark, ctx, False)
    return M, 0


def norm_crop_image(image, landmark, image_size=112, mode="arcface", *, ctx=None):
    """helpers.py:56-59 -> uint8 [112,112,3]."""
    if image_size != 112:
        raise NotImplementedError("only 112x112 crops are implemented on the device")
    crop, _ = _align(image, landmark, ctx, True)
    return crop


def _decode(points, distance, ncol, ctx):
    ctx = _ctx(ctx)
    points = np.ascontiguousarray(points, dtype=np.float32)
    distance = np.ascontiguousarray(distance, dtype=np.float32)
    n = points.shape[0]
    if n == 0:
        return np.zeros((0, ncol), np.float32)
    p, d = ctx.to_device(points), ctx.to_device(distance)
    o = ctx.empty((n, ncol), np.float32)
    if ncol == 4:
        check(ctx.lib.fid_distance2bbox(ctx.handle, C.c_void_p(p.ptr), C.c_void_p(d.ptr), n, C.c_void_p(o.ptr)))
    else:
        check(ctx.lib.fid_distance2kps(ctx.handle, C.c_void_p(p.ptr), C.c_void_p(d.ptr), n, ncol, C.c_void_p(o.ptr)))
    return o.download()


def distance2bbox(points, distance, max_shape=None, *, ctx=None):
    """helpers.py:62-83"""
    if max_shape is not None:
        raise NotImplementedError("max_shape clamping is unused on the reference path")
    return _decode(points, distance, 4, ctx)


def distance2kps(points, distance, max_shape=None, *, ctx=None):
    """helpers.py:86-107"""
    if max_shape is not None:
        raise NotImplementedError("max_shape clamping is unused on the reference path")
    return _decode(points, distance, np.asarray(distance).shape[1], ctx)


def compute_similarity(feat1: np.ndarray, feat2: np.ndarray, *, ctx=None) -> np.float32:
    """helpers.py:110-123: cosine of two feature vectors (np.float32)."""
    idx, score, cos = match_gallery(np.asarray(feat2).reshape(1, -1), np.asarray(feat1).reshape(1, -1), -2.0, ctx=ctx,
                                    return_matrix=True)
    return np.float32(cos[0, 0])


def match_gallery(embeddings, gallery, thresh, *, ctx=None, return_matrix=False):
    """The gallery scan of reference main.py:136-142 for many faces at once.
    embeddings [N,D], gallery [G,D] (raw, un-normalised) -> (idx int32 [N] (-1 = Unknown), score float32 [N])."""
    from ..engine import Gallery
    ctx = _ctx(ctx)
    emb = np.ascontiguousarray(embeddings, dtype=np.float32)
    n, dim = emb.shape
    gal = gallery if isinstance(gallery, Gallery) else Gallery(ctx, np.asarray(gallery, dtype=np.float32))
    e = ctx.to_device(emb)
    q = ctx.empty((n, dim), np.float16)
    check(ctx.lib.fid_l2_normalize_f16(ctx.handle, C.c_void_p(e.ptr), n, dim, C.c_void_p(q.ptr)))
    idx, sc = ctx.empty((n,), np.int32), ctx.empty((n,), np.float32)
    gal.match_device(q, n, thresh, idx, sc)
    out = (idx.download(), sc.download())
    if return_matrix:
        cm = ctx.empty((n, gal.Gp), np.float32)
        check(ctx.lib.fid_cosine_matrix(ctx.handle, gal.handle, C.c_void_p(q.ptr), n, C.c_void_p(cm.ptr)))
        out = out + (cm.download()[:, :gal.G],)
    if not isinstance(gallery, Gallery):
        gal.close()
    return out


# ---- drawing (reference utils/helpers.py:126-179): visualisation only, needs OpenCV on the host ----
def _cv2():
    try:
        import cv2
        return cv2
    except ImportError as e:            # pragma: no cover
        raise ImportError("draw_bbox / draw_bbox_info need opencv-python on the host") from e


def draw_bbox(image, bbox, color=(0, 255, 0), thickness=3, proportion=0.2):
    cv2 = _cv2()
    x1, y1, x2, y2 = map(int, bbox)
    cv2.rectangle(image, (x1, y1), (x2, y2), color, max(1, thickness // 3))
    L = int(min(x2 - x1, y2 - y1) * proportion)
    for (cx, cy, sx, sy) in ((x1, y1, 1, 1), (x2, y1, -1, 1), (x1, y2, 1, -1), (x2, y2, -1, -1)):
        cv2.line(image, (cx, cy), (cx + sx * L, cy), color, thickness)
        cv2.line(image, (cx, cy), (cx, cy + sy * L), color, thickness)
    return image


def draw_bbox_info(frame, bbox, similarity, name, color):
    cv2 = _cv2()
    x1, y1, _, _ = map(int, bbox)
    cv2.putText(frame, f"{name}: {similarity:.2f}", (x1, y1 - 10), cv2.FONT_HERSHEY_SIMPLEX, 1, color, 2)
    draw_bbox(frame, bbox, color)

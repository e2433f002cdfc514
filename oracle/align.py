"""Oracle: 5-point alignment, OpenCV-style fixed-point warp / resize, blob conversion.
Test infrastructure only.

  estimate_norm      reference utils/helpers.py:18-53 + skimage 0.18.3 _umeyama
                     (transform/_geometric.py:72-144), SURVEY.md A.2   [pinned: tests/golden/umeyama.npz]
  warp_affine        cv2.warpAffine as called at utils/helpers.py:58, SURVEY.md A.3   [PARITY UNPINNED]
  resize_linear      cv2.resize as called at models/scrfd.py:135, SURVEY.md A.1       [PARITY UNPINNED]
  blob_from_images   cv2.dnn.blobFromImage(s) at scrfd.py:76-82 / arcface.py:44-50, A.4 [PARITY UNPINNED]
"""
import numpy as np

REFERENCE_ALIGNMENT = np.array(
    [[[38.2946, 51.6963], [73.5318, 51.5014], [56.0252, 71.7366],
      [41.5493, 92.3655], [70.7299, 92.2041]]], dtype=np.float32)   # helpers.py:6-15


def umeyama(src, dst, estimate_scale=True):
    """skimage _umeyama, line by line (dtype follows the inputs exactly like numpy does there)."""
    num, dim = src.shape
    src_mean = src.mean(axis=0)
    dst_mean = dst.mean(axis=0)
    src_demean = src - src_mean
    dst_demean = dst - dst_mean
    A = dst_demean.T @ src_demean / num
    d = np.ones((dim,), dtype=np.double)
    if np.linalg.det(A) < 0:
        d[dim - 1] = -1
    T = np.eye(dim + 1, dtype=np.double)
    U, S, V = np.linalg.svd(A)
    rank = np.linalg.matrix_rank(A)
    if rank == 0:
        return np.nan * T
    elif rank == dim - 1:
        if np.linalg.det(U) * np.linalg.det(V) > 0:
            T[:dim, :dim] = U @ V
        else:
            s = d[dim - 1]
            d[dim - 1] = -1
            T[:dim, :dim] = U @ np.diag(d) @ V
            d[dim - 1] = s
    else:
        T[:dim, :dim] = U @ np.diag(d) @ V
    scale = 1.0 / src_demean.var(axis=0).sum() * (S @ d) if estimate_scale else 1.0
    T[:dim, dim] = dst_mean - scale * (T[:dim, :dim] @ src_mean.T)
    T[:dim, :dim] *= scale
    return T


def estimate_norm(landmark, image_size=112, f64=False):
    """helpers.py:18-53 (single template -> index 0).  f64=True evaluates the same estimate from
    the f32 landmarks in double precision (what the HIP kernel does)."""
    assert landmark.shape == (5, 2)
    alignment = REFERENCE_ALIGNMENT if image_size == 112 else float(image_size) / 112 * REFERENCE_ALIGNMENT
    if f64:
        T = umeyama(landmark.astype(np.float64), alignment[0].astype(np.float64))
    else:
        T = umeyama(landmark, alignment[0])
    return T[0:2, :], 0


def _cv_round(x):
    """cvRound / saturate_cast<int>(double): round half to even (lrint)."""
    return np.rint(x).astype(np.int64)


def warp_affine(image, M, out_size=112):
    """cv2.warpAffine(image, M, (S,S), borderValue=0.0): INTER_LINEAR, BORDER_CONSTANT, u8, 3 ch.
    Fixed-point path of OpenCV 4.x (AB_BITS=10, INTER_BITS=5, 15-bit weights)."""
    H, W = image.shape[:2]
    M = np.asarray(M, dtype=np.float64)
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[1, 1] * D, M[0, 0] * D
    m00, m01, m10, m11 = A11, -M[0, 1] * D, -M[1, 0] * D, A22
    m02 = -m00 * M[0, 2] - m01 * M[1, 2]
    m12 = -m10 * M[0, 2] - m11 * M[1, 2]
    xs = np.arange(out_size, dtype=np.float64)
    adelta = _cv_round(m00 * xs * 1024)
    bdelta = _cv_round(m10 * xs * 1024)
    ys = np.arange(out_size, dtype=np.float64)
    X0 = _cv_round((m01 * ys + m02) * 1024) + 16
    Y0 = _cv_round((m11 * ys + m12) * 1024) + 16
    # saturate_cast<int> of the per-row/per-column terms (int32), sums wrap like C int arithmetic
    i32 = lambda v: np.clip(v, -2**31, 2**31 - 1)
    X = (i32(X0)[:, None] + i32(adelta)[None, :]) >> 5
    Y = (i32(Y0)[:, None] + i32(bdelta)[None, :]) >> 5
    sx = np.clip(X >> 5, -32768, 32767)          # saturate_cast<short>
    sy = np.clip(Y >> 5, -32768, 32767)
    fx = X & 31
    fy = Y & 31
    w00 = (32 - fx) * (32 - fy)
    w01 = fx * (32 - fy)
    w10 = (32 - fx) * fy
    w11 = fx * fy
    img = image.astype(np.int64)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        v = img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]
        return v * ok[..., None]

    acc = (w00[..., None] * tap(sy, sx) + w01[..., None] * tap(sy, sx + 1)
           + w10[..., None] * tap(sy + 1, sx) + w11[..., None] * tap(sy + 1, sx + 1))
    return ((acc + 512) >> 10).astype(np.uint8)


def norm_crop_image(image, landmark, image_size=112):
    """helpers.py:56-59"""
    M, _ = estimate_norm(landmark, image_size, f64=True)
    return warp_affine(image, M, image_size)


def resize_linear(image, new_w, new_h):
    """cv2.resize(image, (new_w, new_h)) default INTER_LINEAR for u8."""
    H, W = image.shape[:2]
    if (new_w, new_h) == (W, H):
        return image.copy()
    if W == 2 * new_w and H == 2 * new_h:      # OpenCV switches exact 2x decimation to INTER_AREA
        v = image.astype(np.int32)
        return ((v[0::2, 0::2] + v[0::2, 1::2] + v[1::2, 0::2] + v[1::2, 1::2] + 2) >> 2).astype(np.uint8)

    def coeffs(src, dst):
        scale = src / dst                                     # double
        d = np.arange(dst, dtype=np.float64)
        f = ((d + 0.5) * scale - 0.5).astype(np.float32)      # (float) cast
        s = np.floor(f).astype(np.int64)
        f = f - s.astype(np.float32)
        lo = s < 0
        s[lo], f[lo] = 0, 0
        hi = s >= src - 1
        s[hi], f[hi] = src - 1, 0
        a1 = np.rint(f * np.float32(2048)).astype(np.int64)
        a0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
        return s, np.minimum(s + 1, src - 1), a0, a1

    sx, sx1, a0, a1 = coeffs(W, new_w)
    sy, sy1, b0, b1 = coeffs(H, new_h)
    src = image.astype(np.int64)
    T = src[:, sx] * a0[None, :, None] + src[:, sx1] * a1[None, :, None]        # [H, new_w, 3]
    T0, T1 = T[sy], T[sy1]
    out = (((b0[:, None, None] * (T0 >> 4)) >> 16) + ((b1[:, None, None] * (T1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def letterbox(image, input_size=(640, 640)):
    """scrfd.py:123-138: returns (det_image u8 [h,w,3], det_scale)."""
    from .postprocess import letterbox_geometry
    width, height = input_size
    new_w, new_h, det_scale = letterbox_geometry(image.shape[0], image.shape[1], input_size)
    det = np.zeros((height, width, 3), dtype=np.uint8)
    det[:new_h, :new_w, :] = resize_linear(image, new_w, new_h)
    return det, det_scale


def blob_from_images(images, scale, mean):
    """cv2.dnn.blobFromImages(images, scale, size, (mean,)*3, swapRB=True) with size == image size:
    blob[n,c,y,x] = (img[n][y,x,2-c] - mean) * scale, float32."""
    x = np.stack([np.asarray(i) for i in images]).astype(np.float32)
    x = x[..., ::-1]
    x = (x - np.float32(mean)) * np.float32(scale)
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2))

"""GPU parity: SCRFD post-process (threshold/decode/sort/NMS/max_num) through the C-ABI, bit-exact
against the reference-generated goldens and the oracle."""
import ctypes as C

import numpy as np
import pytest

from conftest import dense_heads, load_golden
from oracle import postprocess as pp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from scrfd_arcface_facerecognition_amd._lib import Context
    c = Context(0)
    yield c
    c.close()


def run_post(ctx, heads_per_frame, img_hw, max_num=0, metric=0, conf=0.5, iou=0.4, cap=1024, cand_cap=4096):
    from scrfd_arcface_facerecognition_amd.engine import HeadViews, PostProcessor
    B = len(heads_per_frame)
    bufs = [ctx.to_device(np.stack([h[k] for h in heads_per_frame])) for k in range(9)]
    hv = HeadViews.from_onnx_layout(bufs)
    post = PostProcessor(ctx, B, cap=cap, cand_cap=cand_cap)
    post.run(hv, B, (640, 640), img_hw, conf, iou, max_num, metric)
    return post.fetch(B)


def test_detect_goldens_bit_exact(ctx):
    g = load_golden("detect.npz")
    for ci in range(int(g["n_cases"])):
        p = f"c{ci}_"
        outs = dense_heads(g[p + "pos"], g[p + "pos_score"], g[p + "pos_bbox"], g[p + "pos_kps"])
        (det, kps), = run_post(ctx, [outs], tuple(int(v) for v in g[p + "shape"]), int(g[p + "max_num"]), int(g[p + "metric"]))
        assert det.shape == g[p + "det"].shape and kps.shape == g[p + "kps"].shape, ci
        assert np.array_equal(det, g[p + "det"]), ci
        assert np.array_equal(kps, g[p + "kps"]), ci


def test_batched_frames_match_oracle(ctx):
    rng = np.random.default_rng(0)
    frames = []
    for b in range(6):
        K = [0, 3, 40, 400, 1500, 90][b]
        total = 16800
        scores = rng.uniform(0.0, 0.45, total).astype(np.float32)
        pos = rng.choice(total, K, replace=False)
        scores[pos] = rng.permutation(np.linspace(0.5, 0.99, max(K, 1)))[:K].astype(np.float32)
        bbox = rng.uniform(-1, 8, (total, 4)).astype(np.float32)
        kps = rng.uniform(-6, 6, (total, 10)).astype(np.float32)
        ns = [12800, 3200, 800]
        o = np.cumsum([0] + ns)
        frames.append([scores[o[i]:o[i + 1], None] for i in range(3)] + [bbox[o[i]:o[i + 1]] for i in range(3)]
                      + [kps[o[i]:o[i + 1]] for i in range(3)])
    for max_num, metric in ((0, 0), (2, 0), (5, 1)):
        res = run_post(ctx, frames, (1080, 1920), max_num, metric)
        for b, (det, kps) in enumerate(res):
            odet, okps = pp.detect_from_heads(frames[b], (1080, 1920), max_num=max_num, metric="max" if metric == 0 else "d")
            assert np.array_equal(det, odet), (b, max_num)
            assert np.array_equal(kps, okps), (b, max_num)


def test_all_anchors_above_threshold(ctx):
    """maximum size: every one of the 16800 anchors is a candidate (conf_thres = 0)."""
    rng = np.random.default_rng(1)
    total = 16800
    scores = rng.permutation(np.linspace(0.01, 0.99, total)).astype(np.float32)
    bbox = rng.uniform(0.5, 3, (total, 4)).astype(np.float32)
    kps = rng.uniform(-3, 3, (total, 10)).astype(np.float32)
    o = np.cumsum([0, 12800, 3200, 800])
    heads = [scores[o[i]:o[i + 1], None] for i in range(3)] + [bbox[o[i]:o[i + 1]] for i in range(3)] + [kps[o[i]:o[i + 1]] for i in range(3)]
    (det, kps_), = run_post(ctx, [heads], (640, 640), conf=0.0, cap=8192, cand_cap=16800)
    odet, okps = pp.detect_from_heads(heads, (640, 640), conf_thres=0.0)
    assert np.array_equal(det, odet) and np.array_equal(kps_, okps)


def test_candidate_overflow_is_reported(ctx):
    from scrfd_arcface_facerecognition_amd._lib import FaceIdError
    total = 16800
    scores = np.full(total, 0.9, np.float32)
    o = np.cumsum([0, 12800, 3200, 800])
    z4, z10 = np.zeros((total, 4), np.float32), np.zeros((total, 10), np.float32)
    heads = [scores[o[i]:o[i + 1], None] for i in range(3)] + [z4[o[i]:o[i + 1]] for i in range(3)] + [z10[o[i]:o[i + 1]] for i in range(3)]
    with pytest.raises(FaceIdError):
        run_post(ctx, [heads], (640, 640), cand_cap=1024)


def test_nms_goldens(ctx):
    g = load_golden("nms.npz")
    for ci in range(int(g["n_cases"])):
        dets = g[f"c{ci}_dets"]
        K = len(dets)
        d = ctx.to_device(dets)
        keep = ctx.empty((K,), np.int32)
        cnt = ctx.empty((1,), np.int32)
        from scrfd_arcface_facerecognition_amd._lib import check
        check(ctx.lib.fid_nms(ctx.handle, C.c_void_p(d.ptr), K, float(g[f"c{ci}_thr"]), C.c_void_p(keep.ptr), C.c_void_p(cnt.ptr)))
        n = int(cnt.download()[0])
        assert np.array_equal(keep.download()[:n].astype(np.int64), g[f"c{ci}_keep"]), ci


def test_forward_decode_goldens(ctx):
    from scrfd_arcface_facerecognition_amd._lib import check, c_void_pp
    from scrfd_arcface_facerecognition_amd.engine import HeadViews
    g = load_golden("forward.npz")
    for ci in range(int(g["n_cases"])):
        p = f"c{ci}_"
        outs = dense_heads(g[p + "pos"], g[p + "pos_score"], g[p + "pos_bbox"], g[p + "pos_kps"])
        bufs = [ctx.to_device(o[None]) for o in outs]
        hv = HeadViews.from_onnx_layout(bufs)
        rec = ctx.empty((1, 4096, 16), np.float32)
        cnt = ctx.empty((1,), np.int32)
        check(ctx.lib.fid_scrfd_set_candidate_capacity(ctx.handle, 4096))
        check(ctx.lib.fid_scrfd_decode(ctx.handle, C.cast(hv.ptrs, c_void_pp), hv.pix, hv.anc, hv.bstride, 1, 640, 640, 2,
                                       float(g[p + "thr"]), C.c_void_p(rec.ptr), C.c_void_p(cnt.ptr)))
        n = int(cnt.download()[0])
        r = rec.download()[0, :n]
        flat = r[:, 15].view(np.int32)
        lv_of = np.digitize(flat, [12800, 16000])
        for lv in range(3):
            m = lv_of == lv
            assert np.array_equal(r[m, 4:5], g[p + f"scores{lv}"])
            assert np.array_equal(r[m, 0:4], g[p + f"bboxes{lv}"])
            assert np.array_equal(r[m, 5:15].reshape(-1, 5, 2), g[p + f"kpss{lv}"])


def test_tied_scores_follow_the_one_rule(ctx):
    """Duplicate scores inside one stride and across strides (tests/test_oracle_golden.py::tied_heads pins the rule on the oracle:
    detect -> score descending, flat anchor ascending; of two overlapping equal-score candidates the lower anchor survives; max_num
    prefers the later of equal areas; nms() alone walks ties in descending index order): the device must agree bit for bit."""
    from test_oracle_golden import tied_heads
    outs, cand = tied_heads()
    for max_num, metric in ((0, 0), (3, 0), (4, 1)):
        (det, kps), = run_post(ctx, [outs], (640, 640), max_num, metric)
        odet, okps = pp.detect_from_heads(outs, (640, 640), max_num=max_num, metric="max" if metric == 0 else "d")
        assert np.array_equal(det, odet) and np.array_equal(kps, okps), (max_num, metric)
    # a batch whose frames differ only by which candidates tie
    outs2, _ = tied_heads(seed=6, n_groups=2)
    res = run_post(ctx, [outs, outs2, outs], (640, 640))
    for heads, (det, kps) in zip((outs, outs2, outs), res):
        odet, okps = pp.detect_from_heads(heads, (640, 640))
        assert np.array_equal(det, odet) and np.array_equal(kps, okps)
    # fid_nms on a plain det array with tied scores
    from scrfd_arcface_facerecognition_amd._lib import check
    rng = np.random.default_rng(2)
    K = 300
    xy = rng.uniform(0, 600, (K, 2)).astype(np.float32)
    wh = rng.uniform(10, 60, (K, 2)).astype(np.float32)
    dets = np.concatenate([xy, xy + wh, rng.choice(np.float32([0.9, 0.8, 0.7, 0.6]), (K, 1))], axis=1).astype(np.float32)
    d = ctx.to_device(dets)
    keep, cnt = ctx.empty((K,), np.int32), ctx.empty((1,), np.int32)
    check(ctx.lib.fid_nms(ctx.handle, C.c_void_p(d.ptr), K, 0.4, C.c_void_p(keep.ptr), C.c_void_p(cnt.ptr)))
    n = int(cnt.download()[0])
    assert [int(k) for k in keep.download()[:n]] == [int(k) for k in pp.nms(dets, 0.4)]

#!/usr/bin/env python3
"""Generate golden vectors for the hot path from the reference's own Python code.

Runs ONLY in the build container (needs --reference /root/reference, which does not
exist on the GPU box).  The reference cannot be imported as-is here because `cv2`,
`onnxruntime` and `skimage` are not installed, so inert stub modules are registered
first.  The stubs contain NO arithmetic of the functions under
test: `cv2.resize` returns a zero image of the requested size, `blobFromImage` returns
a zero blob (the fake session ignores it), drawing calls are no-ops.  Everything that
ends up in the fixtures is therefore computed by reference code alone:

  models/scrfd.py:70-207     forward (threshold+decode), detect (sort/NMS/max_num), nms
  utils/helpers.py:62-123    distance2bbox, distance2kps, compute_similarity
  main.py:108-150            frame_processor gallery scan (strict >, first max wins)

`estimate_norm` (utils/helpers.py:18-53) needs the real skimage; that part is produced by
tools/gen_golden_umeyama.py under /opt/conda/bin/python3.9 (skimage 0.18.3).

Outputs: tests/golden/*.npz  (data only: inputs + expected outputs).
"""
import argparse
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True


def install_stubs():
    cv2 = types.ModuleType("cv2")

    def resize(img, dsize, *a, **k):
        w, h = dsize
        return np.zeros((h, w, 3), dtype=np.uint8)

    def blob_from_image(img, scale, size, mean, swapRB=False):
        w, h = size
        return np.zeros((1, 3, h, w), dtype=np.float32)

    def noop(*a, **k):
        return None

    cv2.resize = resize
    cv2.dnn = types.SimpleNamespace(blobFromImage=blob_from_image, blobFromImages=noop)
    for n in ("rectangle", "line", "putText", "imread", "imshow", "waitKey", "VideoCapture",
              "VideoWriter", "VideoWriter_fourcc", "destroyAllWindows", "warpAffine", "circle"):
        setattr(cv2, n, noop)
    cv2.getTextSize = lambda *a, **k: ((10, 10), 2)
    cv2.FONT_HERSHEY_SIMPLEX = 0
    cv2.FILLED = -1
    cv2.LINE_AA = 16
    cv2.CAP_PROP_FRAME_WIDTH = 3
    cv2.CAP_PROP_FRAME_HEIGHT = 4
    cv2.CAP_PROP_FPS = 5
    sys.modules["cv2"] = cv2
    ort = types.ModuleType("onnxruntime")
    ort.InferenceSession = object
    sys.modules["onnxruntime"] = ort
    sk = types.ModuleType("skimage")
    skt = types.ModuleType("skimage.transform")
    skt.SimilarityTransform = object
    sk.transform = skt
    sys.modules["skimage"] = sk
    sys.modules["skimage.transform"] = skt


class FakeSession:
    """Stands in for onnxruntime.InferenceSession.run: returns the 9 head tensors."""

    def __init__(self, outs):
        self.outs = outs

    def run(self, names, feed):
        return [o.copy() for o in self.outs]


def make_detector(SCRFD, outs, conf=0.5, iou=0.4, input_size=(640, 640)):
    d = SCRFD.__new__(SCRFD)
    d.input_size = input_size
    d.conf_thres = conf
    d.iou_thres = iou
    d.fmc = 3
    d._feat_stride_fpn = [8, 16, 32]
    d._num_anchors = 2
    d.use_kps = True
    d.mean = 127.5
    d.std = 128.0
    d.center_cache = {}
    d.session = FakeSession(outs)
    d.input_names = ["input.1"]
    d.output_names = [str(i) for i in range(9)]
    return d


def synth_heads(rng, K, size=640, clustered=True):
    """Fake SCRFD head tensors with exactly K anchors >= 0.5 and tie-free scores."""
    ns = [(size // s) * (size // s) * 2 for s in (8, 16, 32)]
    total = sum(ns)
    scores = np.full(total, 0.125, dtype=np.float32)      # constant background (< any threshold used)
    if K > 0:
        if clustered and K >= 4:
            # clusters of neighbouring anchors so that NMS has real work to do
            centers = rng.choice(total - 8, size=max(1, K // 4), replace=False)
            pos = np.unique(np.concatenate([centers + d for d in range(4)]))[:K]
            if len(pos) < K:
                rest = np.setdiff1d(np.arange(total), pos)
                pos = np.concatenate([pos, rng.choice(rest, K - len(pos), replace=False)])
        else:
            pos = rng.choice(total, size=K, replace=False)
        vals = np.linspace(0.5, 0.999, num=4 * K + 7, dtype=np.float64)
        vals = rng.choice(vals, size=K, replace=False).astype(np.float32)
        assert len(np.unique(vals)) == K
        scores[pos] = vals
    bbox = np.zeros((total, 4), dtype=np.float32)
    kps = np.zeros((total, 10), dtype=np.float32)
    if K > 0:
        bbox[pos] = rng.uniform(0.3, 6.0, size=(K, 4)).astype(np.float32)
        kps[pos] = rng.uniform(-4.0, 4.0, size=(K, 10)).astype(np.float32)
    outs, o = [], 0
    for n in ns:
        outs.append(scores[o:o + n].reshape(n, 1))
        o += n
    o = 0
    for n in ns:
        outs.append(bbox[o:o + n])
        o += n
    o = 0
    for n in ns:
        outs.append(kps[o:o + n])
        o += n
    return outs


def sparse_heads(outs):
    """Fixtures store only the anchors that differ from the constant background."""
    scores = np.concatenate([o.ravel() for o in outs[0:3]])
    bbox = np.vstack(outs[3:6])
    kps = np.vstack(outs[6:9])
    pos = np.nonzero(scores != np.float32(0.125))[0].astype(np.int32)
    return pos, scores[pos], bbox[pos], kps[pos]


def gen_detect(SCRFD, out_dir):
    rng = np.random.default_rng(20251004)
    cases = {}
    idx = 0
    shapes = [(640, 640), (1080, 1920), (1280, 1280), (480, 853)]
    for K in (0, 1, 5, 20, 200, 700):
        for (H, W) in shapes if K in (20, 200) else shapes[:2]:
            outs = synth_heads(rng, K)
            img = np.zeros((H, W, 3), dtype=np.uint8)
            for max_num, metric in ((0, "max"), (1, "max"), (3, "max"), (3, "default")):
                d = make_detector(SCRFD, outs)
                det, kpss = d.detect(img, max_num=max_num, metric=metric)
                pre = f"c{idx}_"
                cases[pre + "shape"] = np.array([H, W], dtype=np.int32)
                cases[pre + "max_num"] = np.array(max_num, dtype=np.int32)
                cases[pre + "metric"] = np.array(0 if metric == "max" else 1, dtype=np.int32)
                pos, ps, pb, pk = sparse_heads(outs)
                cases[pre + "pos"], cases[pre + "pos_score"] = pos, ps
                cases[pre + "pos_bbox"], cases[pre + "pos_kps"] = pb, pk
                cases[pre + "det"] = det
                cases[pre + "kps"] = kpss
                assert det.dtype == np.float32 and kpss.dtype == np.float32
                idx += 1
    cases["n_cases"] = np.array(idx, dtype=np.int32)
    np.savez_compressed(os.path.join(out_dir, "detect.npz"), **cases)
    print("detect cases:", idx)


def gen_forward(SCRFD, out_dir):
    rng = np.random.default_rng(7)
    cases = {}
    for ci, (K, thr) in enumerate(((50, 0.5), (300, 0.5), (50, 0.3), (0, 0.5))):
        outs = synth_heads(rng, K, clustered=False)
        d = make_detector(SCRFD, outs, conf=thr)
        img = np.zeros((640, 640, 3), dtype=np.uint8)
        s, b, k = d.forward(img, thr)
        pre = f"c{ci}_"
        cases[pre + "thr"] = np.array(thr, dtype=np.float64)
        pos, ps, pb, pk = sparse_heads(outs)
        cases[pre + "pos"], cases[pre + "pos_score"] = pos, ps
        cases[pre + "pos_bbox"], cases[pre + "pos_kps"] = pb, pk
        for lv in range(3):
            cases[pre + f"scores{lv}"] = s[lv]
            cases[pre + f"bboxes{lv}"] = b[lv]
            cases[pre + f"kpss{lv}"] = k[lv]
    cases["n_cases"] = np.array(4, dtype=np.int32)
    # anchor-centre order (scrfd.py:102-105), read back from the reference's cache
    d = make_detector(SCRFD, synth_heads(rng, 3))
    d.forward(np.zeros((640, 640, 3), np.uint8), 0.5)
    for (h, w, s), v in d.center_cache.items():
        cases[f"centers_s{s}"] = v
    np.savez_compressed(os.path.join(out_dir, "forward.npz"), **cases)


def gen_nms(SCRFD, out_dir):
    rng = np.random.default_rng(11)
    d = make_detector(SCRFD, synth_heads(rng, 0))
    cases = {}
    ci = 0

    def add(dets, thr):
        nonlocal ci
        keep = d.nms(dets, iou_thres=thr)
        cases[f"c{ci}_dets"] = dets
        cases[f"c{ci}_thr"] = np.array(thr, dtype=np.float64)
        cases[f"c{ci}_keep"] = np.array(keep, dtype=np.int64)
        ci += 1

    for K in (1, 2, 20, 200, 1000):
        xy = rng.uniform(0, 600, size=(K, 2)).astype(np.float32)
        wh = rng.uniform(5, 120, size=(K, 2)).astype(np.float32)
        sc = rng.permutation(np.linspace(0.5, 0.99, K)).astype(np.float32)
        dets = np.hstack([xy, xy + wh, sc[:, None]]).astype(np.float32)
        add(dets, 0.4)
        add(dets, 0.1)
    # dense cluster: heavy suppression
    K = 300
    base = np.array([100, 100, 200, 220], dtype=np.float32)
    jit = rng.uniform(-30, 30, size=(K, 4)).astype(np.float32)
    sc = rng.permutation(np.linspace(0.5, 0.99, K)).astype(np.float32)
    add(np.hstack([base + jit, sc[:, None]]).astype(np.float32), 0.4)
    # adversarial: integer boxes whose IoU (with the +1 convention) is exactly 0.4 = 2/5
    # box A = [0,0,9,9] area 100; B = [0,0,9,3] area 40 -> inter 40, union 100 -> 0.4 (kept, <=)
    dets = np.array([[0, 0, 9, 9, 0.9], [0, 0, 9, 3, 0.8], [0, 0, 9, 4, 0.7],
                     [20, 20, 29, 29, 0.6], [20, 20, 29, 23, 0.55]], dtype=np.float32)
    add(dets, 0.4)
    # degenerate / negative-size boxes (random weights produce them)
    dets = np.array([[10, 10, 5, 5, 0.9], [0, 0, 50, 50, 0.8], [4, 4, 12, 12, 0.7],
                     [10, 10, 5, 5, 0.6], [-5, -5, 3, 3, 0.5]], dtype=np.float32)
    add(dets, 0.4)
    cases["n_cases"] = np.array(ci, dtype=np.int32)
    np.savez_compressed(os.path.join(out_dir, "nms.npz"), **cases)
    print("nms cases:", ci)


def gen_decode_helpers(H, out_dir):
    rng = np.random.default_rng(3)
    pts = rng.uniform(0, 640, size=(500, 2)).astype(np.float32)
    dist = rng.uniform(-10, 50, size=(500, 4)).astype(np.float32)
    kd = rng.uniform(-30, 30, size=(500, 10)).astype(np.float32)
    np.savez_compressed(os.path.join(out_dir, "decode.npz"), points=pts, dist=dist, kdist=kd,
                        bbox=H.distance2bbox(pts, dist), kps=H.distance2kps(pts, kd))


def gen_cosine(H, out_dir):
    rng = np.random.default_rng(5)
    a = rng.standard_normal((64, 512)).astype(np.float32)
    b = rng.standard_normal((64, 512)).astype(np.float32)
    b[0] = a[0]                       # identical
    b[1] = -a[1]                      # opposite
    b[2] = 0
    b[2, :256] = a[2, 256:]           # unrelated halves
    a[3] *= 1e-3
    b[3] = a[3] * 7.0                 # scale invariance
    sims = np.array([H.compute_similarity(x, y) for x, y in zip(a, b)])
    assert sims.dtype == np.float32
    np.savez_compressed(os.path.join(out_dir, "cosine.npz"), a=a, b=b, sim=sims)


def gen_gallery_scan(main_mod, out_dir):
    """Drive the reference frame_processor (main.py:108-150) with fake detector/recognizer and
    record which (name, similarity) it reports per face."""
    rng = np.random.default_rng(9)
    G, N = 40, 24
    gallery = rng.standard_normal((G, 512)).astype(np.float32)
    emb = rng.standard_normal((N, 512)).astype(np.float32)
    emb[0] = gallery[7] * 3.0                       # exact match -> sim 1
    emb[1] = gallery[9] + 0.8 * rng.standard_normal(512).astype(np.float32)
    emb[2] = -gallery[3]                            # all sims <= 0 for that entry
    emb[3] = gallery[5] + gallery[6]                # near tie between two entries
    gallery[21] = gallery[20]                       # duplicated entry: first max wins (strict >)
    emb[4] = gallery[20] * 0.5
    for i in range(5, 12):
        j = int(rng.integers(0, G))
        emb[i] = gallery[j] + rng.uniform(0.5, 2.5) * rng.standard_normal(512).astype(np.float32)
    targets = [(gallery[j], f"id{j}") for j in range(G)]
    results = {}
    for thr in (0.4, 0.0, 0.9):
        rec = []

        class Det:
            def detect(self, frame, max_num=0):
                return (np.tile(np.array([[1, 2, 3, 4, 0.9]], np.float32), (N, 1)),
                        np.zeros((N, 5, 2), np.float32))

        class Rec:
            def __init__(self):
                self.i = 0

            def __call__(self, frame, kps):
                e = emb[self.i]
                self.i += 1
                return e

        main_mod.draw_bbox_info = lambda frame, bbox, similarity, name, color: rec.append((name, similarity))
        main_mod.draw_bbox = lambda frame, bbox, color: rec.append(("Unknown", 0))
        params = types.SimpleNamespace(max_num=0, similarity_thresh=thr)
        colors = {n: (0, 0, 0) for _, n in targets}
        main_mod.frame_processor(np.zeros((8, 8, 3), np.uint8), Det(), Rec(), targets, colors, params)
        assert len(rec) == N
        idx = np.array([-1 if n == "Unknown" else int(n[2:]) for n, _ in rec], dtype=np.int32)
        sim = np.array([float(s) for _, s in rec], dtype=np.float32)
        results[f"idx_thr{thr}"] = idx
        results[f"sim_thr{thr}"] = sim
    np.savez_compressed(os.path.join(out_dir, "gallery_scan.npz"), gallery=gallery, emb=emb,
                        thrs=np.array([0.4, 0.0, 0.9]), **results)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    args = ap.parse_args()
    if not os.path.isdir(args.reference):
        sys.exit("reference tree not present: fixtures can only be regenerated in the build container")
    install_stubs()
    sys.path.insert(0, args.reference)
    import utils.helpers as H          # reference module
    from models.scrfd import SCRFD     # reference class
    import main as ref_main            # reference driver
    os.makedirs(args.out, exist_ok=True)
    gen_decode_helpers(H, args.out)
    gen_forward(SCRFD, args.out)
    gen_detect(SCRFD, args.out)
    gen_nms(SCRFD, args.out)
    gen_cosine(H, args.out)
    gen_gallery_scan(ref_main, args.out)
    print("wrote fixtures to", os.path.abspath(args.out))


if __name__ == "__main__":
    main()

"""The oracle against the vectors produced by the reference's own code (tools/gen_golden*.py).
CPU only.  This is what pins the oracle."""
import numpy as np

from conftest import dense_heads, load_golden
from oracle import align, match
from oracle import postprocess as pp


def test_decode_helpers_bit_exact():
    g = load_golden("decode.npz")
    assert np.array_equal(pp.distance2bbox(g["points"], g["dist"]), g["bbox"])
    assert np.array_equal(pp.distance2kps(g["points"], g["kdist"]), g["kps"])


def test_anchor_centers_order():
    g = load_golden("forward.npz")
    for s in (8, 16, 32):
        c = pp.anchor_centers(640 // s, 640 // s, s)
        assert c.dtype == np.float32 and np.array_equal(c, g[f"centers_s{s}"])
    assert pp.anchor_centers(80, 80, 8)[:5].tolist() == [[0, 0], [0, 0], [8, 0], [8, 0], [16, 0]]


def test_forward_decode_bit_exact():
    g = load_golden("forward.npz")
    for ci in range(int(g["n_cases"])):
        p = f"c{ci}_"
        outs = dense_heads(g[p + "pos"], g[p + "pos_score"], g[p + "pos_bbox"], g[p + "pos_kps"])
        s, b, k = pp.decode_heads(outs, (640, 640), float(g[p + "thr"]))
        for lv in range(3):
            assert np.array_equal(s[lv], g[p + f"scores{lv}"])
            assert np.array_equal(b[lv], g[p + f"bboxes{lv}"])
            assert np.array_equal(k[lv], g[p + f"kpss{lv}"])


def test_detect_bit_exact():
    g = load_golden("detect.npz")
    n = int(g["n_cases"])
    assert n == 64
    seen_empty = False
    for ci in range(n):
        p = f"c{ci}_"
        outs = dense_heads(g[p + "pos"], g[p + "pos_score"], g[p + "pos_bbox"], g[p + "pos_kps"])
        det, kps = pp.detect_from_heads(outs, tuple(g[p + "shape"]), max_num=int(g[p + "max_num"]),
                                        metric="max" if int(g[p + "metric"]) == 0 else "default")
        assert det.shape == g[p + "det"].shape and kps.shape == g[p + "kps"].shape
        assert det.dtype == np.float32 and kps.dtype == np.float32
        assert np.array_equal(det, g[p + "det"]), ci
        assert np.array_equal(kps, g[p + "kps"]), ci
        seen_empty |= det.shape == (0, 5) and kps.shape == (0, 5, 2)
    assert seen_empty


def test_nms_bit_exact():
    g = load_golden("nms.npz")
    for ci in range(int(g["n_cases"])):
        keep = pp.nms(g[f"c{ci}_dets"], float(g[f"c{ci}_thr"]))
        assert np.array_equal(np.asarray(keep, dtype=np.int64), g[f"c{ci}_keep"]), ci


def test_nms_iou_exactly_at_threshold_is_kept():
    dets = np.array([[0, 0, 9, 9, 0.9], [0, 0, 9, 3, 0.8], [0, 0, 9, 4, 0.7]], dtype=np.float32)
    assert pp.nms(dets, 0.4) == [0, 1]            # IoU 0.4 kept (<=), 0.5 suppressed


def test_estimate_norm_vs_skimage():
    g = load_golden("umeyama.npz")
    for lm, M in zip(g["landmarks"], g["M"]):
        scale = max(1.0, np.abs(M).max())
        m32, idx = align.estimate_norm(lm)
        assert idx == 0 and m32.dtype == np.float64 and m32.shape == (2, 3)
        assert np.abs(m32 - M).max() / scale < 2e-6          # same algorithm, LAPACK build differs
        m64, _ = align.estimate_norm(lm, f64=True)
        assert np.abs(m64 - M).max() / scale < 1e-5          # <= 6.6e-6 expected


def test_cosine_vs_reference():
    g = load_golden("cosine.npz")
    sims = np.array([match.compute_similarity(a, b) for a, b in zip(g["a"], g["b"])])
    assert sims.dtype == np.float32
    # np.dot summation order is BLAS dependent: equal to a few ulp
    assert np.abs(sims - g["sim"]).max() < 1e-6
    assert abs(sims[0] - 1) < 1e-6 and abs(sims[1] + 1) < 1e-6 and abs(sims[3] - 1) < 1e-6


def test_gallery_scan_semantics():
    g = load_golden("gallery_scan.npz")
    for thr in g["thrs"]:
        idx = g[f"idx_thr{thr}"]
        sim = g[f"sim_thr{thr}"]
        for i, e in enumerate(g["emb"]):
            j, s = match.gallery_scan(e, g["gallery"], float(thr))
            assert j == idx[i], (thr, i)
            assert abs(float(s) - float(sim[i])) < 1e-6
        bj, bs = match.match_batch(g["emb"], g["gallery"], float(thr))
        assert np.array_equal(bj, idx)
        assert np.abs(bs - sim).max() < 1e-5
    assert g["idx_thr0.4"][0] == 7 and g["idx_thr0.4"][2] == -1 and g["idx_thr0.4"][4] == 20


def test_crop_bytes_f32_vs_f64_transform():
    """a13's real "bit-exact vs reference" error bar.  skimage's estimate runs in fp32 on fp32 landmarks
    (the pinned goldens, tests/golden/umeyama.npz); the oracle's and the device's crops use the same closed form in fp64
    (M differs by <= 6.6e-6 relative).  Count the crop bytes that change between warp_affine(golden fp32-derived M) and
    warp_affine(fp64 M) over all 129 golden landmark sets on a random frame: the fixed-point warp quantises source
    coordinates to 1/1024 px (A.3), so only coordinates that sit within ~1e-5 px of a rounding boundary can flip.
    The bound asserted here is the one DESIGN.md section 5 quotes."""
    from oracle import align
    g = load_golden("umeyama.npz")
    rng = np.random.default_rng(0)
    frame = rng.integers(0, 256, (720, 1280, 3), dtype=np.uint8)
    total = flipped = worst = faces_changed = 0
    max_step = 0
    for lm, M32 in zip(g["landmarks"], g["M"]):
        M64, _ = align.estimate_norm(lm, f64=True)
        a = align.warp_affine(frame, M32)
        b = align.warp_affine(frame, M64)
        d = a != b
        n = int(d.sum())
        total += a.size
        flipped += n
        worst = max(worst, n)
        faces_changed += n > 0
        if n:
            max_step = max(max_step, int(np.abs(a.astype(np.int16) - b.astype(np.int16)).max()))
    frac = flipped / total
    print(f"crop bytes differing between the fp32-derived and the fp64 transform: {flipped} of {total} ({frac:.2e}); "
          f"worst face {worst} of 37632 bytes; {faces_changed} of {len(g['M'])} faces touched; largest step {max_step}")
    assert frac < 2e-3 and worst < 1200


def tied_heads(seed=5, n_groups=6):
    """Dense SCRFD heads (640x640, 2 anchors) whose candidates hold DUPLICATED scores -- inside one stride and across strides
    (saturated sigmoid outputs of fp16 towers do tie) -- with boxes far enough apart that NMS keeps every one of them, plus two
    tied candidates that overlap (one must suppress the other).  Returns (outs[9], flat anchor ids of the candidates)."""
    rng = np.random.default_rng(seed)
    ns = [12800, 3200, 800]
    total = sum(ns)
    scores = np.full(total, 0.125, dtype=np.float32)
    bbox = np.zeros((total, 4), dtype=np.float32)
    kps = np.zeros((total, 10), dtype=np.float32)
    # anchors on a coarse grid of the stride-8 map (pixel pitch 80 = 640 px / 8 cells: boxes of ~24 px never touch), both anchors of
    # some cells, and anchors of the stride-16 / stride-32 maps at other places
    cells8 = [(y * 80 + x) * 2 for y in range(5, 75, 10) for x in range(5, 75, 10)]
    cand = list(rng.choice(cells8, 20, replace=False)) + [12800 + (y * 40 + x) * 2 for y, x in ((3, 3), (3, 30), (30, 3))] \
        + [16000 + (y * 20 + x) * 2 + 1 for y, x in ((1, 18), (18, 1))]
    vals = np.float32([0.984375, 0.75, 0.99951171875, 0.5625, 0.875, 0.6875])[:n_groups]
    for i, a in enumerate(cand):
        scores[a] = vals[i % len(vals)]                          # every value is shared by 4+ candidates, across strides
        bbox[a] = 1.5
        kps[a] = rng.uniform(-1, 1, 10)
    a0 = cand[0]
    scores[a0 + 1] = scores[a0]                                  # the other anchor of the same cell: same centre, same box -> IoU 1
    bbox[a0 + 1] = 1.5
    cand.append(a0 + 1)
    outs, o = [], 0
    for n in ns:
        outs.append(scores[o:o + n].reshape(n, 1)); o += n
    o = 0
    for n in ns:
        outs.append(bbox[o:o + n]); o += n
    o = 0
    for n in ns:
        outs.append(kps[o:o + n]); o += n
    return outs, np.array(sorted(cand))


def test_tie_order_rule_of_the_oracle():
    """ONE tie rule, pinned: the reference sorts twice with numpy's default argsort (scrfd.py:144 in detect,
    :188 inside nms), which is unstable -- the oracle makes both sorts stable, and a stable ascending sort reversed, twice, leaves
    equal scores in ASCENDING flat-anchor order when NMS walks them.  So detect() returns (score descending, anchor ascending) and of two
    overlapping candidates with equal scores the LOWER anchor survives; SCRFD.nms() alone (one reversal) walks ties in DESCENDING
    index order; max_num's argsort (one reversal) prefers the LATER of two equal areas."""
    outs, cand = tied_heads()
    det, kps = pp.detect_from_heads(outs, (640, 640))
    flat_scores = np.concatenate([o.ravel() for o in outs[:3]])
    want = sorted((int(a) for a in cand), key=lambda a: (-float(flat_scores[a]), a))
    dropped = int(cand[np.flatnonzero(np.diff(cand) == 1)[0] + 1])      # the second anchor of the doubled cell loses to the first
    want.remove(dropped)
    assert len(det) == len(want)
    assert np.array_equal(det[:, 4], flat_scores[want])
    # identify every returned row by its decoded box centre -> anchor
    ns = [12800, 3200, 800]
    centres = []
    for a in want:
        lv = 0 if a < 12800 else (1 if a < 16000 else 2)
        s = (8, 16, 32)[lv]
        pix = (a - (0, 12800, 16000)[lv]) // 2
        centres.append(((pix % (640 // s)) * s, (pix // (640 // s)) * s))
    got = [((r[0] + r[2]) / 2, (r[1] + r[3]) / 2) for r in det]
    assert np.allclose(got, centres)
    # nms() alone: ties in descending index order
    d = np.array([[0, 0, 10, 10, .9], [100, 0, 110, 10, .9], [200, 0, 210, 10, .9], [0, 0, 10, 10, .5]], dtype=np.float32)
    assert [int(k) for k in pp.nms(d, 0.4)] == [2, 1, 0]
    # max_num with equal areas: the later one first
    det2, _ = pp.detect_from_heads(outs, (640, 640), max_num=3)
    area = (det[:, 2] - det[:, 0]) * (det[:, 3] - det[:, 1])
    pick = sorted(range(len(det)), key=lambda i: (-float(area[i]), -i))[:3]
    assert len(set(np.round(area[pick], 3))) < 3                      # (the selection did have to break a tie)
    assert np.array_equal(det2, det[pick])


def test_face_gates_match_the_reference():
    """oracle/gates.py against the reference's own assess_face_quality / is_side_face / analyze_bbox_for_side_face
    (smart_face_recognition.py:1145-1399; tools/gen_golden_gates.py, 600 faces incl. the threshold values): bit for bit"""
    import json
    from oracle import gates
    g = load_golden("gates.npz")
    cfg = gates.config_from_reference_json(json.loads(bytes(g["config"]).decode()))
    assert cfg == gates.DEFAULT_CONFIG                      # the defaults ARE the reference's config.json
    for i in range(len(g["score"])):
        b = g["bbox"][i]
        assert np.array_equal(gates.face_quality(b, g["kps"][i], g["score"][i], cfg).astype(np.float64), g["quality"][i]), i
        flag, score = gates.bbox_side_score(b[2] - b[0], b[3] - b[1], b[1], b[0], g["score"][i], cfg)
        assert (int(flag), score) == tuple(g["bbox_side"][i]), i
        assert int(gates.is_side_face(b, g["score"][i], g["pose"][i, 0], g["pose"][i, 1], cfg)) == g["side"][i], i
    # best-face selection (:1473-1519): first maximum, rejections in the reference's order
    dets = np.array([[100, 100, 200, 220, 0.7], [300, 80, 420, 230, 0.9], [50, 300, 170, 440, 0.9]], np.float32)
    kps = np.tile(np.array([[120, 130], [180, 130], [150, 160], [125, 190], [175, 190]], np.float32), (3, 1, 1))
    assert gates.select_best(dets, kps)[:2] == (1, gates.ACCEPT)
    assert gates.select_best(dets[:0], kps[:0])[:2] == (-1, gates.NO_FACE)
    low = dets.copy(); low[:, 4] = (0.3, 0.59, 0.2)
    assert gates.select_best(low, kps)[:2] == (1, gates.LOW_CONFIDENCE)
    side = dets.copy(); side[1, :4] = (300, 80, 325, 230)      # ratio 0.17: extreme profile
    assert gates.select_best(side, kps)[:2] == (1, gates.SIDE_FACE)
    assert gates.select_best(dets, kps, dict(cfg, min_quality_threshold=0.99))[:2] == (1, gates.LOW_QUALITY)
    assert gates.select_best(dets, kps, cfg, poses=[(0, 0), (np.radians(50.0), 0), (0, 0)])[:2] == (1, gates.SIDE_FACE)


def test_product_layer_duplicate_logic_restatement():
    """oracle/match.py's restatement of the product layer's store logic (reference smart_face_recognition.py:2632-2641, 2726-2797,
    qdrant_manager.py:137-183; Qdrant itself unpinned) on a hand-checkable store: ids 5 ~ 9 ~ 12 form a chain (5 !~ 12), 7 ~ 20 a pair."""
    from oracle import match
    e = np.zeros((5, 4), np.float32)
    ids = [9, 5, 12, 20, 7]

    def unit(a):
        return np.array([np.cos(a), np.sin(a), 0, 0], np.float32)
    e[1], e[0], e[2] = unit(0.0), unit(0.5), unit(1.0)               # cos(0.5) = 0.878 >= 0.8, cos(1.0) = 0.54 < 0.8
    e[4], e[3] = np.array([0, 0, 1, 0], np.float32), np.array([0, 0, 0.99, 0.14], np.float32)
    hits = match.search_similar(unit(0.1), ids, e, k=2, threshold=0.8)
    assert [h[0] for h in hits] == [5, 9]
    assert match.is_duplicate_embedding(unit(0.01), ids, e, 0.95) and not match.is_duplicate_embedding(unit(1.6), ids, e, 0.95)       # cos(0.6) = 0.825 to the nearest
    merges, alive = match.find_and_merge_duplicates(ids, e, 0.8)
    # 5 absorbs 9 (0.878); 12 was within reach of 9 only, and 9 is gone when 12's turn comes: it survives; 7 absorbs 20
    assert [(a, b) for a, b, _ in merges] == [(5, 9), (7, 20)] and alive == [5, 7, 12]

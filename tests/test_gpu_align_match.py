"""GPU parity: alignment (Umeyama + fixed-point warp, letterbox resize) and gallery match."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_golden
from oracle import align, match

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from scrfd_arcface_facerecognition_amd._lib import Context
    c = Context(0)
    yield c
    c.close()


def align_on_gpu(ctx, frames, kps, counts, F):
    from scrfd_arcface_facerecognition_amd._lib import check
    B, H, W, _ = frames.shape
    cap = kps.shape[1]
    fr, kp, cn = ctx.to_device(frames), ctx.to_device(kps.astype(np.float32)), ctx.to_device(counts.astype(np.int32))
    crops = ctx.empty((B * F, 112, 112, 3), np.uint8)
    M = ctx.empty((B * F, 6), np.float64)
    check(ctx.lib.fid_align_crops(ctx.handle, C.c_void_p(fr.ptr), B, H, W, C.c_void_p(kp.ptr), C.c_void_p(cn.ptr), cap, F,
                                  C.c_void_p(crops.ptr), C.c_void_p(M.ptr)))
    return crops.download(), M.download().reshape(B * F, 2, 3)


def test_estimate_norm_vs_skimage_goldens(ctx):
    g = load_golden("umeyama.npz")
    lms = g["landmarks"]
    n = len(lms)
    frames = np.zeros((1, 8, 8, 3), np.uint8)
    _, M = align_on_gpu(ctx, frames, lms.reshape(1, n, 10), np.array([n]), n)
    for i in range(n):
        scale = max(1.0, np.abs(g["M"][i]).max())
        assert np.abs(M[i] - g["M"][i]).max() / scale < 1e-5, i       # SURVEY.md A.2 tolerance (skimage runs fp32 SVD)
        m64, _ = align.estimate_norm(lms[i], f64=True)                # same closed form in fp64: ~1e-12
        assert np.abs(M[i] - m64).max() / scale < 1e-9, i


def test_warp_bit_exact_vs_oracle(ctx):
    rng = np.random.default_rng(3)
    B, H, W, F = 3, 360, 480, 4
    frames = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    tmpl = align.REFERENCE_ALIGNMENT[0].astype(np.float64)
    kps = np.zeros((B, F, 10), np.float32)
    for b in range(B):
        for f in range(F):
            s, th = rng.uniform(0.4, 3.0), rng.uniform(-1.0, 1.0)
            R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
            t = np.array([rng.uniform(-40, W + 40), rng.uniform(-40, H + 40)])   # some faces hang over the border
            kps[b, f] = ((tmpl - 56) @ R.T * s + t + rng.normal(0, 1.0, (5, 2))).reshape(-1)
    counts = np.array([4, 2, 0])
    crops, M = align_on_gpu(ctx, frames, kps, counts, F)
    crops = crops.reshape(B, F, 112, 112, 3)
    for b in range(B):
        for f in range(F):
            if f < counts[b]:
                ref = align.norm_crop_image(frames[b], kps[b, f].reshape(5, 2))
                assert np.array_equal(crops[b, f], ref), (b, f)
                assert ref.max() > 0
            else:
                assert not crops[b, f].any()


@pytest.mark.parametrize("shape", [(1080, 1920), (480, 853), (1280, 1280), (640, 640), (700, 500)])
def test_letterbox_bit_exact_vs_oracle(ctx, shape):
    from scrfd_arcface_facerecognition_amd._lib import check
    rng = np.random.default_rng(5)
    H, W = shape
    frames = rng.integers(0, 256, (2, H, W, 3), dtype=np.uint8)
    fr = ctx.to_device(frames)
    out = ctx.empty((2, 640, 640, 3), np.uint8)
    sc = C.c_double()
    check(ctx.lib.fid_letterbox(ctx.handle, C.c_void_p(fr.ptr), 2, H, W, C.c_void_p(out.ptr), 640, 640, C.byref(sc)))
    got = out.download()
    for b in range(2):
        ref, scale = align.letterbox(frames[b])
        assert scale == sc.value
        assert np.array_equal(got[b], ref), shape


def gpu_match(ctx, emb, gallery, thr):
    from scrfd_arcface_facerecognition_amd._lib import check
    from scrfd_arcface_facerecognition_amd.engine import Gallery
    gal = Gallery(ctx, gallery)
    n, dim = emb.shape
    e = ctx.to_device(emb.astype(np.float32))
    q = ctx.empty((n, dim), np.float16)
    check(ctx.lib.fid_l2_normalize_f16(ctx.handle, C.c_void_p(e.ptr), n, dim, C.c_void_p(q.ptr)))
    idx, sc = ctx.empty((n,), np.int32), ctx.empty((n,), np.float32)
    gal.match_device(q, n, thr, idx, sc)
    cm = ctx.empty((n, gal.Gp), np.float32)
    check(ctx.lib.fid_cosine_matrix(ctx.handle, gal.handle, C.c_void_p(q.ptr), n, C.c_void_p(cm.ptr)))
    res = idx.download(), sc.download(), cm.download()[:, :gal.G]
    gal.close()
    return res


def test_gallery_scan_goldens(ctx):
    g = load_golden("gallery_scan.npz")
    for thr in g["thrs"]:
        idx, sc, cm = gpu_match(ctx, g["emb"], g["gallery"], float(thr))
        ref_idx, ref_sim = g[f"idx_thr{thr}"], g[f"sim_thr{thr}"]
        e = g["emb"] / np.linalg.norm(g["emb"], axis=1, keepdims=True)
        gg = g["gallery"] / np.linalg.norm(g["gallery"], axis=1, keepdims=True)
        true = e @ gg.T
        for i in range(len(idx)):
            if idx[i] != ref_idx[i]:
                # fp16 unit vectors: only entries whose true cosine is within 1e-3 of the winner
                # (rows 3/4 are deliberate twins) or of the threshold may be swapped
                if idx[i] >= 0 and ref_idx[i] >= 0:
                    assert abs(true[i, idx[i]] - true[i, ref_idx[i]]) < 1e-3, (thr, i)
                else:
                    assert abs(true[i].max() - max(float(thr), 0.0)) < 1e-3, (thr, i)
                continue
            assert abs(sc[i] - ref_sim[i]) < 1e-3, (thr, i)        # north_star tolerance for cosine in fp16


def test_cosine_matrix_within_1e3(ctx):
    g = load_golden("cosine.npz")
    _, _, cm = gpu_match(ctx, g["a"], g["b"], 0.0)
    got = np.diag(cm)
    ok = np.isfinite(g["sim"])
    assert np.abs(got[ok] - g["sim"][ok]).max() < 1e-3


@pytest.mark.parametrize("n,G", [(1, 5), (64, 1000), (300, 4097), (7, 33)])
def test_match_random_vs_oracle(ctx, n, G):
    rng = np.random.default_rng(n * 1000 + G)
    gallery = rng.standard_normal((G, 512)).astype(np.float32)
    emb = rng.standard_normal((n, 512)).astype(np.float32)
    for i in range(0, n, 2):       # half of the queries are noisy copies of a gallery row
        emb[i] = gallery[rng.integers(0, G)] + 0.7 * rng.standard_normal(512).astype(np.float32)
    idx, sc, cm = gpu_match(ctx, emb, gallery, 0.4)
    oi, osim = match.match_batch(emb, gallery, 0.4)
    e = emb / np.linalg.norm(emb, axis=1, keepdims=True)
    gg = gallery / np.linalg.norm(gallery, axis=1, keepdims=True)
    assert np.abs(cm - e @ gg.T).max() < 1e-3
    margin = np.abs(np.sort(e @ gg.T, axis=1)[:, -1] - 0.4) > 2e-3     # decisions not within fp16 noise of the threshold
    assert np.array_equal(idx[margin], oi[margin])
    assert np.abs(sc[margin] - osim[margin]).max() < 1e-3


def test_vector_gallery_topk_upsert_delete(ctx):
    """top-k search + upsert/delete (the QdrantManager analogue) against a numpy reference."""
    from scrfd_arcface_facerecognition_amd.engine import VectorGallery
    rng = np.random.default_rng(21)
    vg = VectorGallery(ctx, 512, capacity=64)
    emb = {f"p{i}": rng.standard_normal(512).astype(np.float32) for i in range(150)}      # forces two capacity doublings
    ids = list(emb)
    vg.upsert(ids[:100], np.stack([emb[i] for i in ids[:100]]))
    vg.upsert(ids[100:], np.stack([emb[i] for i in ids[100:]]))
    vg.delete(ids[10:20])
    emb["p3"] = rng.standard_normal(512).astype(np.float32)
    vg.upsert(["p3"], emb["p3"][None])                                                    # overwrite in place
    live = [i for i in ids if i not in ids[10:20]]
    assert len(vg) == len(live) == 140
    M = np.stack([emb[i] for i in live]); M /= np.linalg.norm(M, axis=1, keepdims=True)
    queries = np.stack([emb["p3"] + 0.3 * rng.standard_normal(512), emb["p120"], emb["p15"], rng.standard_normal(512)]).astype(np.float32)
    qn = queries / np.linalg.norm(queries, axis=1, keepdims=True)
    true = qn @ M.T
    for k in (1, 5, 8):
        res = vg.search(queries, k=k, score_threshold=0.05)
        for r, hits in enumerate(res):
            order = np.argsort(-true[r])
            want = [(live[j], true[r, j]) for j in order[:k] if true[r, j] > 0.05]
            assert len(hits) == len(want), (k, r)
            for (hid, hs), (wid, ws) in zip(hits, want):
                assert abs(hs - ws) < 1e-3
                assert hid == wid or abs(true[r, live.index(hid)] - ws) < 1e-3
    assert vg.search(queries[:2], k=1)[0][0][0] == "p3" and vg.search(queries[:2], k=1)[1][0][0] == "p120"
    assert all(h[0] != "p15" for h in vg.search(queries[2:3], k=8, score_threshold=0.0)[0])   # deleted id never returned


def _duplicate_store(rng, n=120):
    """person embeddings with planted duplicate structure: tight clusters (everyone within 0.8 of everyone), a CHAIN a ~ b ~ c whose ends are
    below the threshold (the greedy order decides who absorbs whom), near-copies (>= 0.95) and unrelated persons"""
    base = rng.standard_normal((n, 512)).astype(np.float32)

    def near(v, cos):                                        # a vector at cosine `cos` from v
        v = v / np.linalg.norm(v)
        r = rng.standard_normal(512).astype(np.float32)
        r -= (r @ v) * v
        r /= np.linalg.norm(r)
        return (cos * v + np.sqrt(1 - cos * cos) * r).astype(np.float32) * np.float32(rng.uniform(0.5, 2.0))
    for i, j, c in ((40, 3, 0.93), (77, 3, 0.90), (78, 40, 0.97), (15, 90, 0.86), (91, 15, 0.99), (60, 61, 0.96)):
        base[i] = near(base[j], c)
    base[100] = near(base[50], 0.88)                         # chain: 50 ~ 100 ~ 101, 50 !~ 101
    v50 = base[50] / np.linalg.norm(base[50])
    v100 = base[100] / np.linalg.norm(base[100])
    away = v100 - (v100 @ v50) * v50
    away /= np.linalg.norm(away)
    base[101] = (0.88 * v100 + np.sqrt(1 - 0.88 ** 2) * (0.9 * away + np.sqrt(1 - 0.81) * near(away, 0.0) / np.linalg.norm(near(away, 0.0)))).astype(np.float32)
    return base


def test_vector_gallery_duplicate_check_and_merge(ctx):
    """SURVEY 8 f-3, the product layer's use of the store: `is_duplicate` (config.json duplicate_similarity_threshold 0.95) and
    `find_and_merge_duplicates` (merge_duplicate_threshold 0.8) against the oracle's restatement of smart_face_recognition.py:2632-2641 / 2726-2797
    (the Qdrant search itself is unpinned: no qdrant_client offline).  Pairs within 2e-3 of the threshold are kept out of the fixture by construction
    and checked for; ids are inserted in a shuffled order so that store rows and id order differ."""
    from scrfd_arcface_facerecognition_amd.engine import VectorGallery
    rng = np.random.default_rng(55)
    emb = _duplicate_store(rng)
    ids = [int(i) for i in rng.permutation(1000)[:len(emb)]]
    unit = emb / np.linalg.norm(emb, axis=1, keepdims=True)
    S = unit @ unit.T
    assert np.abs(np.abs(S[np.triu_indices(len(emb), 1)]) - 0.8).min() > 2e-3 and np.abs(S[np.triu_indices(len(emb), 1)] - 0.95).min() > 2e-3
    vg = VectorGallery(ctx, 512, capacity=64)
    order = rng.permutation(len(ids))
    vg.upsert([ids[j] for j in order], emb[order])
    # duplicate check of new embeddings: a near copy, a moderately similar one, an unrelated one
    probes = np.stack([emb[7] * 3.0 + 0.01 * rng.standard_normal(512), emb[7] + 6.0 * rng.standard_normal(512), rng.standard_normal(512)]).astype(np.float32)
    for p in probes:
        assert vg.is_duplicate(p) == match.is_duplicate_embedding(p, ids, emb, 0.95)
    assert vg.is_duplicate(probes[0]) and not vg.is_duplicate(probes[2])
    want, survivors = match.find_and_merge_duplicates(ids, emb, 0.8)
    got = vg.find_and_merge_duplicates(0.8)
    assert len(want) >= 7 and [(a, b) for a, b, _ in got] == [(a, b) for a, b, _ in want]
    assert max(abs(g[2] - w[2]) for g, w in zip(got, want)) < 1e-3
    assert sorted(vg.row_of) == survivors and len(vg) == len(ids) - len(want)
    assert vg.find_and_merge_duplicates(0.8) == []                    # idempotent: nothing left above the threshold
    # the chain: the smaller id of (50's, 100's) absorbs the other; 101's person survives iff it was not within reach of the absorber
    kept = {a for a, _, _ in want} | set(survivors)
    assert all(b not in kept for _, b, _ in want)


def test_zero_gallery_row_does_not_poison_the_match(ctx):
    """an all-zero target used to become a NaN fp16 row whose key outranked every real score and turned EVERY
    query into Unknown.  The reference skips such a target (`nan > x` is False, main.py:139-140) and matches the others."""
    rng = np.random.default_rng(17)
    G, n = 300, 40
    gallery = rng.standard_normal((G, 512)).astype(np.float32)
    gallery[0] = 0.0
    gallery[131] = 0.0
    emb = rng.standard_normal((n, 512)).astype(np.float32)
    tgt = rng.integers(1, G, n)
    tgt[tgt == 131] = 132
    for i in range(0, n, 2):
        emb[i] = gallery[tgt[i]] + 0.3 * rng.standard_normal(512).astype(np.float32)
    emb[3] = 0.0                                                       # an all-zero query as well
    idx, sc, cm = gpu_match(ctx, emb, gallery, 0.4)
    assert np.isfinite(cm).all() and not cm[:, 0].any() and not cm[:, 131].any() and not cm[3].any()
    oi, osim = match.match_batch(emb, gallery, 0.4)
    assert np.array_equal(idx, oi) and np.abs(sc - osim).max() < 1e-3
    for i in range(0, n, 2):
        j, s = match.gallery_scan(emb[i], gallery, 0.4)               # the reference-structured python loop itself
        assert idx[i] == j == tgt[i] and abs(sc[i] - s) < 1e-3
    assert idx[3] == -1 and sc[3] == 0.0


def test_match_on_vector_gallery_free_rows(ctx):
    """fid_match on a VectorGallery's inner gallery: its free capacity rows are zero rows (never NaN) and never match."""
    from scrfd_arcface_facerecognition_amd._lib import check
    from scrfd_arcface_facerecognition_amd.engine import VectorGallery
    rng = np.random.default_rng(18)
    vg = VectorGallery(ctx, 512, capacity=64)
    emb = rng.standard_normal((10, 512)).astype(np.float32)
    vg.upsert([f"p{i}" for i in range(10)], emb)
    vg.delete(["p4"])
    rows = ctx.borrow(__import__("scrfd_arcface_facerecognition_amd.engine", fromlist=["_gallery_ptr"])._gallery_ptr(vg._gal),
                      (vg._gal.Gp, 512), np.float16).download()
    assert np.isfinite(rows.astype(np.float32)).all()
    queries = np.concatenate([emb + 0.2 * rng.standard_normal((10, 512)).astype(np.float32), rng.standard_normal((3, 512)).astype(np.float32)])
    n = len(queries)
    e = ctx.to_device(queries)
    q = ctx.empty((n, 512), np.float16)
    check(ctx.lib.fid_l2_normalize_f16(ctx.handle, C.c_void_p(e.ptr), n, 512, C.c_void_p(q.ptr)))
    idx, sc = ctx.empty((n,), np.int32), ctx.empty((n,), np.float32)
    vg._gal.match_device(q, n, 0.4, idx, sc)
    idx, sc = idx.download(), sc.download()
    for i in range(10):
        if i == 4:
            assert idx[i] == -1
        else:
            assert vg.id_of[int(idx[i])] == f"p{i}" and sc[i] > 0.9
    assert (idx[10:] == -1).all()

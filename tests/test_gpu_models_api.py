"""GPU: the mirrored call surface (models.SCRFD / models.ArcFace / utils.helpers) behaves like the
reference's, and the batched pipeline agrees with the frame-by-frame oracle."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import align as oalign
from oracle import match as omatch
from oracle import nets as onets
from oracle import pipeline as opipe
from oracle import postprocess as pp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from scrfd_arcface_facerecognition_amd._lib import default_context
    return default_context(0)


@pytest.fixture(scope="module")
def detector():
    from models import SCRFD          # the reference's import path (main.py:11)
    return SCRFD("synthetic:scrfd_2.5g?seed=3", input_size=(640, 640), conf_thres=0.5)


@pytest.fixture(scope="module")
def recognizer():
    from models import ArcFace
    return ArcFace("synthetic:arcface_mbf?seed=3")


def test_scrfd_attributes_and_empty_result(detector):
    d = detector
    assert d.input_size == (640, 640) and d.conf_thres == 0.5 and d.iou_thres == 0.4
    assert d.fmc == 3 and d._feat_stride_fpn == [8, 16, 32] and d._num_anchors == 2 and d.use_kps
    assert d.mean == 127.5 and d.std == 128.0 and d.center_cache == {}
    assert len(d.output_names) == 9 and len(d.input_names) == 1
    d.conf_thres = 2.0                                          # unreachable score: the K = 0 path
    try:
        det, kps = d.detect(np.zeros((480, 640, 3), np.uint8))
    finally:
        d.conf_thres = 0.5
    assert det.shape == (0, 5) and kps.shape == (0, 5, 2) and det.dtype == np.float32 and kps.dtype == np.float32


def test_missing_model_raises_like_the_reference(capsys):
    from models import SCRFD
    with pytest.raises(FileNotFoundError):
        SCRFD("./weights/det_10g.onnx")
    assert "Failed to load the model" in capsys.readouterr().out


def test_detect_equals_oracle_on_the_gpu_heads(detector, ctx):
    """detect() = letterbox + net + post-process.  The decisions are checked bit-exactly by feeding the
    oracle's post-process the SAME head tensors the GPU net produced (fp16 nets cannot be bit-equal to
    fp32 ones; head closeness is tested in test_gpu_nets)."""
    from scrfd_arcface_facerecognition_amd.pipeline import calibrate_detector_bias
    from models import SCRFD
    rng = np.random.default_rng(5)
    frame = rng.integers(0, 256, (720, 1280, 3), dtype=np.uint8)
    det_img, _ = oalign.letterbox(frame)
    P, _ = calibrate_detector_bias(ctx, detector.session.net, detector.session.params, det_img[None], target=60, max_batch=1)
    d = SCRFD.__new__(SCRFD)
    d.__dict__.update(detector.__dict__)
    from scrfd_arcface_facerecognition_amd.session import HipSession
    d.session = HipSession(None, ctx=ctx, net=detector.session.net, params=P, max_batch=2)
    d._post = None
    heads = d.session.run_images(det_img[None])
    assert [h.shape for h in heads[:3]] == [(12800, 1), (3200, 1), (800, 1)]
    for max_num, metric in ((0, "max"), (1, "max"), (4, "default")):
        det, kps = d.detect(frame, max_num=max_num, metric=metric)
        odet, okps = pp.detect_from_heads(heads, frame.shape[:2], max_num=max_num, metric=metric)
        assert len(odet) > 0
        assert np.array_equal(det, odet) and np.array_equal(kps, okps)
    s, b, k = d.forward(det_img, 0.5)
    os_, ob, ok = pp.decode_heads(heads, (640, 640), 0.5)
    for lv in range(3):
        assert np.array_equal(s[lv], os_[lv]) and np.array_equal(b[lv], ob[lv]) and np.array_equal(k[lv], ok[lv])
    # session.run with the float blob the reference builds gives the same heads
    blob = oalign.blob_from_images([det_img], 1 / 128.0, 127.5)
    heads2 = d.session.run(d.output_names, {d.input_names[0]: blob})
    assert all(np.array_equal(a, b_) for a, b_ in zip(heads, heads2))
    # batched detect ~ per-frame detect: a different batch size picks other conv tiles / split-K plans, so the
    # fp32 summation order and hence single fp16 roundings differ (scores move by ~1e-3, like GPU vs oracle);
    # confident detections must agree
    both = d.detect_batch(np.stack([frame, frame[::-1].copy()]))
    for got, img in ((both[0][0], frame), (both[1][0], frame[::-1].copy())):
        one = d.detect(img)[0]
        strong = one[one[:, 4] > 0.55]
        assert len(strong) > 0
        for row in strong:
            dist = np.abs(got[:, :4] - row[:4]).max(axis=1)
            j = int(dist.argmin())
            assert dist[j] < 1.0 and abs(got[j, 4] - row[4]) < 5e-3


def test_nms_api(detector):
    g = load_golden("nms.npz")
    keep = detector.nms(g["c4_dets"], float(g["c4_thr"]))
    assert isinstance(keep, list) and isinstance(keep[0], np.int64)
    assert np.array_equal(np.asarray(keep), g["c4_keep"])
    assert detector.nms(np.zeros((0, 5), np.float32), 0.4) == []


def test_arcface_surface(recognizer):
    r = recognizer
    assert r.input_mean == 127.5 and r.input_std == 127.5 and r.taskname == "recognition"
    assert r.input_size == (112, 112) and len(r.output_names) == 1
    rng = np.random.default_rng(2)
    frame = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    kps = np.array([[300, 200], [380, 205], [338, 250], [305, 290], [372, 295]], np.float32)
    emb = r(frame, kps)
    assert emb.shape == (512,) and emb.dtype == np.float32
    ref, crop = opipe.embed(frame, kps, r.session.net, r.session.params)
    assert np.array_equal(r.align(frame, kps), crop)
    cos = float(emb @ ref / np.linalg.norm(emb) / np.linalg.norm(ref))
    assert 1 - cos < 1e-3
    feats = r.get_feat([crop, crop[::-1].copy()])
    assert feats.shape == (2, 512) and np.allclose(feats[0], emb, atol=1e-6)
    assert r.get_feat(crop).shape == (1, 512)
    # an injected session (reference arcface.py:11-21) is honoured
    from models import ArcFace
    r2 = ArcFace(session=r.session)
    assert np.array_equal(r2(frame, kps), emb)


def test_helpers_surface():
    from utils.helpers import (compute_similarity, distance2bbox, distance2kps, estimate_norm, norm_crop_image,
                               reference_alignment)
    assert reference_alignment.shape == (1, 5, 2) and reference_alignment.dtype == np.float32
    g = load_golden("decode.npz")
    assert np.array_equal(distance2bbox(g["points"], g["dist"]), g["bbox"])
    assert np.array_equal(distance2kps(g["points"], g["kdist"]), g["kps"])
    gu = load_golden("umeyama.npz")
    M, idx = estimate_norm(gu["landmarks"][-1])
    assert idx == 0 and M.shape == (2, 3) and M.dtype == np.float64
    assert np.abs(M - gu["M"][-1]).max() < 1e-4
    gc = load_golden("cosine.npz")
    s = compute_similarity(gc["a"][5], gc["b"][5])
    assert isinstance(s, np.float32) and abs(s - gc["sim"][5]) < 1e-3
    img = np.random.default_rng(0).integers(0, 256, (200, 300, 3), dtype=np.uint8)
    lm = gu["landmarks"][-1] * 0.5
    assert np.array_equal(norm_crop_image(img, lm), oalign.norm_crop_image(img, lm))


def test_pipeline_matches_frame_by_frame_oracle(ctx):
    """4 frames through the batched device pipeline vs the reference-structured oracle (frame by frame,
    face by face, python gallery loop).  Detector decisions are taken from the GPU heads (see above);
    embeddings / similarities must agree within 1e-3."""
    from scrfd_arcface_facerecognition_amd import archs
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet, Gallery
    from scrfd_arcface_facerecognition_amd.pipeline import FacePipeline, calibrate_detector_bias
    rng = np.random.default_rng(8)
    B, F = 4, 2
    frames = rng.integers(0, 256, (B, 360, 640, 3), dtype=np.uint8)
    det_net = archs.scrfd_500m((640, 640))
    lb = np.stack([oalign.letterbox(f)[0] for f in frames])
    det_P, _ = calibrate_detector_bias(ctx, det_net, archs.synth_params(det_net, 1), lb, target=40, max_batch=4)
    rec_net = archs.mobilefacenet()
    rec_P = archs.synth_params(rec_net, 1)
    gal = rng.standard_normal((37, 512)).astype(np.float32)
    det = CompiledNet(ctx, det_net, det_P, max_batch=B)
    rec = CompiledNet(ctx, rec_net, rec_P, max_batch=B * F)
    pipe = FacePipeline(ctx, det, rec, batch=B, faces_per_frame=F)
    gallery = Gallery(ctx, gal)
    pipe.run_step(ctx.to_device(frames), 360, 640, gallery, thresh=0.02)
    res = pipe.results(gallery)
    emb = pipe.embeddings().reshape(B, F, 512)
    from scrfd_arcface_facerecognition_amd.engine import HeadViews
    for b in range(B):
        heads = []
        for part in range(3):
            for name in det.low.outputs:
                h = det.low.heads[name]
                off, c = (h["score"], h["bbox"], h["kps"])[part]
                fused = det.read(name, B)[b]
                heads.append(np.ascontiguousarray(fused[..., off:off + 2 * c]).reshape(-1, c))
        odet, okps = pp.detect_from_heads(heads, (360, 640), max_num=F)
        assert len(res[b]) == len(odet) == F
        for f in range(F):
            bbox, score, kps, name, sim = res[b][f]
            assert np.array_equal(bbox, odet[f, :4]) and score == odet[f, 4] and np.array_equal(kps, okps[f])
            ref, _ = opipe.embed(frames[b], okps[f], rec_net, rec_P)
            cos = float(ref @ emb[b, f] / np.linalg.norm(ref) / np.linalg.norm(emb[b, f]))
            assert 1 - cos < 1e-3
            j, s = omatch.gallery_scan(ref, gal, 0.02)
            assert abs(s - sim) < 2e-3
            if abs(s - 0.02) > 3e-3:
                e = ref / np.linalg.norm(ref)
                sims = (gal / np.linalg.norm(gal, axis=1, keepdims=True)) @ e
                top2 = np.sort(sims)[-2:]
                if top2[1] - top2[0] > 3e-3:
                    assert name == (gallery.names[j] if j >= 0 else "Unknown")


def test_grouped_pipeline_equals_per_step_pipeline(ctx):
    """GroupedFacePipeline (round 5: the recogniser runs once per `group` steps on their crops): three steps of DIFFERENT frames through a
    group of 2 -- the full group, then a partial group finished by flush() -- give per step exactly the plain FacePipeline's detections and
    identities (same kernels on the same crops; embeddings within the fp16 plan-to-plan tolerance, the batch size may change a kernel pick)."""
    from scrfd_arcface_facerecognition_amd import archs
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet, Gallery
    from scrfd_arcface_facerecognition_amd.pipeline import FacePipeline, GroupedFacePipeline, calibrate_detector_bias
    rng = np.random.default_rng(18)
    B, F, G = 3, 2, 2
    steps = [rng.integers(0, 256, (B, 320, 320, 3), dtype=np.uint8) for _ in range(3)]
    steps[1][1] = 0                                                   # a frame without a face inside the group
    det_net = archs.scrfd_500m((320, 320))
    det_P, _ = calibrate_detector_bias(ctx, det_net, archs.synth_params(det_net, 2), steps[0], target=30, max_batch=B)
    rec_net = archs.mobilefacenet()
    rec_P = archs.synth_params(rec_net, 2)
    gal = rng.standard_normal((29, 512)).astype(np.float32)
    det = CompiledNet(ctx, det_net, det_P, max_batch=B)
    rec = CompiledNet(ctx, rec_net, rec_P, max_batch=G * B * F)
    gallery = Gallery(ctx, gal)
    plain = FacePipeline(ctx, det, rec, batch=B, faces_per_frame=F)
    ref = []
    for fr in steps:
        plain.run_step(ctx.to_device(fr), 320, 320, gallery, thresh=0.02)
        ref.append((plain.results(gallery), plain.q.download().copy()))
    grp = GroupedFacePipeline(ctx, det, rec, batch=B, faces_per_frame=F, group=G)
    dev = [ctx.to_device(fr) for fr in steps]
    grp.run_step(dev[0], 320, 320, gallery, thresh=0.02)
    assert grp.k == 1                                                 # collected, nothing embedded yet
    grp.run_step(dev[1], 320, 320, gallery, thresh=0.02)
    assert grp.k == 0
    got = grp.results(gallery)
    q = grp.q.download()
    grp.run_step(dev[2], 320, 320, gallery, thresh=0.02)
    grp.flush(gallery, thresh=0.02)
    got += grp.results(gallery)
    q = np.concatenate([q, grp.q.download()[:B * F]])
    assert len(got) == 3
    for s in range(3):
        assert len(got[s]) == B
        for b in range(B):
            assert len(got[s][b]) == len(ref[s][0][b])
            for (bb, sc, kp, name, sim), (rbb, rsc, rkp, rname, rsim) in zip(got[s][b], ref[s][0][b]):
                assert np.array_equal(bb, rbb) and sc == rsc and np.array_equal(kp, rkp)
                assert abs(sim - rsim) < 2e-3 and (name == rname or abs(rsim - 0.02) < 3e-3)
        qs, qr = q[s * B * F:(s + 1) * B * F].astype(np.float32), ref[s][1].astype(np.float32)
        assert np.array_equal(qs == 0, qr == 0)                      # the same empty / degenerate slots (incl. the -0.0 marker's magnitude)
        assert np.abs(qs - qr).max() < 2e-3
    assert len(got[1][1]) == 0


def test_build_targets_matches_reference_semantics(ctx, tmp_path, caplog):
    """a19, reference main.py:78-105: per gallery image detect(max_num=1) -> skip + warn when no face -> embed the best face
    -> (embedding, name) with name = filename[:-4], in listing order.  Checked against the frame-by-frame oracle
    (detector decisions from the GPU heads, see test_detect_equals_oracle_on_the_gpu_heads), both for images in memory
    and for a directory (files written as .npy / .ppm, the formats the loader reads without OpenCV)."""
    import logging
    from models import SCRFD, ArcFace
    from scrfd_arcface_facerecognition_amd.pipeline import (build_targets, build_targets_from_images, calibrate_detector_bias,
                                                            gallery_from_targets)
    from scrfd_arcface_facerecognition_amd.session import HipSession
    from scrfd_arcface_facerecognition_amd import archs
    rng = np.random.default_rng(31)
    n_img = 6
    images = [rng.integers(0, 256, (320, 320, 3), dtype=np.uint8) for _ in range(n_img)]
    images[2] = np.zeros((320, 320, 3), np.uint8)
    images[4] = np.full((320, 320, 3), 255, np.uint8)
    names = ["alice", "bob", "blank", "carol", "white", "dave"]
    det_net = archs.scrfd_500m((320, 320))
    det_P, _ = calibrate_detector_bias(ctx, det_net, archs.synth_params(det_net, 5), np.stack([images[i] for i in (0, 1, 3, 5)]),
                                       target=30, max_batch=4)
    # fp32 oracle scores decide which images have a face at all: put conf_thres into the widest gap of the per-image maxima
    blob = oalign.blob_from_images(images, det_net.in_scale, det_net.in_mean)
    mx = []
    for b in range(n_img):
        outs = onets.scrfd_session_outputs(det_net, det_P, blob[b:b + 1])
        mx.append(max(float(o.max()) for o in outs[:3]))
    srt = np.sort(mx)
    gaps = srt[1:] - srt[:-1]
    k = int(np.argmax(gaps))
    assert gaps[k] > 0.02, mx
    thr = float((srt[k] + srt[k + 1]) / 2)
    has_face = [m > thr for m in mx]
    assert any(has_face) and not all(has_face), mx                    # at least one image is skipped
    detector = SCRFD("synthetic:scrfd_500m?seed=5", input_size=(320, 320), conf_thres=thr, max_batch=8)
    detector.session = HipSession(None, ctx=detector.ctx, net=det_net, params=det_P, max_batch=8)
    recognizer = ArcFace("synthetic:arcface_mbf?seed=5")
    rec_net, rec_P = recognizer.session.net, recognizer.session.params
    with caplog.at_level(logging.WARNING):
        targets = build_targets_from_images(detector, recognizer, images, names)
    assert [t[1] for t in targets] == [nm for nm, h in zip(names, has_face) if h]
    assert sum("No face detected" in r.getMessage() for r in caplog.records) == has_face.count(False)
    # oracle per image, landmarks from the GPU heads of the batched run (the compiled net still holds that batch)
    cn = detector.session.compiled((320, 320))
    ti = 0
    for b in range(n_img):
        if not has_face[b]:
            continue
        heads = []
        for part in range(3):
            for name in cn.low.outputs:
                h = cn.low.heads[name]
                off, c = (h["score"], h["bbox"], h["kps"])[part]
                heads.append(np.ascontiguousarray(cn.read(name, n_img)[b][..., off:off + 2 * c]).reshape(-1, c))
        odet, okps = pp.detect_from_heads(heads, (320, 320), (320, 320), thr, 0.4, 1, "max")
        assert len(okps) == 1
        ref, _ = opipe.embed(images[b], okps[0], rec_net, rec_P)
        e = targets[ti][0]
        assert e.shape == (512,) and e.dtype == np.float32
        assert 1 - float(ref @ e / np.linalg.norm(ref) / np.linalg.norm(e)) < 1e-3
        # the per-image reference call sequence gives the same vector as the batched path (batch-1 vs batch-n kernels: fp16 noise)
        _, kpss = detector.detect(images[b], max_num=1)
        # (another batch size -> other autotuned kernels -> another fp32 summation order in the heads: landmarks agree to fp16 noise of
        # the head tensors, 3e-2 stride units in test_gpu_nets.py = well below a quarter pixel at these strides, not bit for bit)
        assert np.abs(kpss[0] - okps[0]).max() < 0.25
        e1 = recognizer(images[b], kpss[0])
        assert 1 - float(e1 @ e / np.linalg.norm(e1) / np.linalg.norm(e)) < 1e-3
        ti += 1
    assert ti == len(targets)
    # directory form: name = filename[:-4]; unreadable files are skipped
    d = tmp_path / "faces"
    d.mkdir()
    for nm, im in zip(names, images):
        if nm in ("bob", "white"):
            with open(d / f"{nm}.ppm", "wb") as f:
                f.write(b"P6\n# synthetic\n320 320\n255\n" + np.ascontiguousarray(im[..., ::-1]).tobytes())
        else:
            np.save(d / f"{nm}.npy", im)
    (d / "notes.txt").write_text("not an image")
    t2 = build_targets(detector, recognizer, str(d))
    by_name = {nm: e for e, nm in t2}
    assert sorted(by_name) == sorted(t[1] for t in targets)
    for e, nm in targets:
        assert np.array_equal(by_name[nm], e)
    gal = gallery_from_targets(ctx, targets)
    assert gal.G == len(targets) and gal.names == [t[1] for t in targets]
    from utils.helpers import match_gallery
    idx, sc = match_gallery(np.stack([t[0] for t in targets]), gal, 0.4)
    assert np.array_equal(idx, np.arange(len(targets))) and np.abs(sc - 1).max() < 1e-3
    gal.close()


def test_one_context_from_two_threads(ctx, monkeypatch):
    """b4 (SURVEY 8b threading; the reference's product layer calls one shared model from <= 4 worker threads,
    smart_face_recognition.py:1954-1957): two Python threads drive ONE fid_ctx (and a second ctx of its own) through
    fid_net_run / fid_match concurrently; every result equals the serial run bit for bit."""
    import threading
    from scrfd_arcface_facerecognition_amd import archs
    from scrfd_arcface_facerecognition_amd._lib import Context
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet, Gallery
    monkeypatch.setenv("FID_AUTOTUNE", "0")       # all four nets on the same (heuristic) kernel plans -> same summation order
    rng = np.random.default_rng(41)
    net = archs.mobilefacenet()
    P = archs.synth_params(net, 4)
    crops = [rng.integers(0, 256, (4, 112, 112, 3), dtype=np.uint8) for _ in range(2)]
    gal_host = rng.standard_normal((700, 512)).astype(np.float32)
    other = Context(0)

    def setup(c):
        return [CompiledNet(c, net, P, max_batch=4) for _ in range(2)], Gallery(c, gal_host)

    def work(c, cn, gal, imgs, out, reps):
        from utils.helpers import match_gallery
        for _ in range(reps):
            cn.run(imgs)
            e = cn.read(cn.low.outputs[0], len(imgs)).reshape(len(imgs), -1)
            idx, sc = match_gallery(e, gal, 0.0, ctx=c)
            out.append((e.copy(), idx.copy(), sc.copy()))

    nets_a, gal_a = setup(ctx)
    nets_b, gal_b = setup(other)
    serial = []
    for t in range(2):
        o = []
        work(ctx, nets_a[t], gal_a, crops[t], o, 1)
        serial.append(o[0])
    outs = [[], [], [], []]
    threads = [threading.Thread(target=work, args=(ctx, nets_a[0], gal_a, crops[0], outs[0], 6)),
               threading.Thread(target=work, args=(ctx, nets_a[1], gal_a, crops[1], outs[1], 6)),       # same ctx, other thread
               threading.Thread(target=work, args=(other, nets_b[0], gal_b, crops[0], outs[2], 6)),     # an independent ctx
               threading.Thread(target=work, args=(other, nets_b[1], gal_b, crops[1], outs[3], 6))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    for k, o in enumerate(outs):
        assert len(o) == 6
        for e, idx, sc in o:
            assert np.array_equal(e, serial[k % 2][0]) and np.array_equal(idx, serial[k % 2][1]) and np.array_equal(sc, serial[k % 2][2])
    other.close()

"""GPU: the FaceAnalysis-style front end and the double-buffered video runner (f-2) give the same
faces as the mirrored per-frame API."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_face_analysis_get_matches_per_face_api():
    from scrfd_arcface_facerecognition_amd.app import FaceAnalysis
    from scrfd_arcface_facerecognition_amd.pipeline import calibrate_detector_bias
    from scrfd_arcface_facerecognition_amd.session import HipSession
    app = FaceAnalysis("synthetic:scrfd_2.5g?seed=1", "synthetic:arcface_mbf?seed=1", max_faces=16)
    frame = np.random.default_rng(1).integers(0, 256, (480, 640, 3), dtype=np.uint8)
    from oracle import align as oalign
    P, _ = calibrate_detector_bias(app.ctx, app.det.session.net, app.det.session.params, oalign.letterbox(frame)[0][None],
                                   target=30, max_batch=1)
    app.det.session = HipSession(None, ctx=app.ctx, net=app.det.session.net, params=P, max_batch=2)
    faces = app.get(frame, max_num=5)
    assert 1 <= len(faces) <= 5
    det, kpss = app.det.detect(frame, max_num=5)
    for i, f in enumerate(faces):
        assert np.array_equal(f.bbox, det[i, :4]) and f.det_score == det[i, 4] and np.array_equal(f.kps, kpss[i])
        e = app.rec(frame, kpss[i])                                   # the reference's per-face call
        assert np.allclose(f.embedding, e, rtol=0, atol=2e-2 * np.abs(e).max())   # batch-1 vs batch-n kernels: fp16 noise
        assert abs(np.linalg.norm(f.normed_embedding) - 1) < 2e-3
        assert np.abs(f.normed_embedding - f.embedding / np.linalg.norm(f.embedding)).max() < 1e-3
    # quality scores / side-face flag of every face and the best-face verdict (fid_face_gates) against the oracle's restatement of
    # smart_face_recognition.py:1145-1216,1248-1399,1473-1519
    from oracle import gates as ogates
    for f in faces:
        q = ogates.face_quality(f.bbox, f.kps, np.float32(f.det_score))
        assert np.array_equal(np.array([f.quality[k] for k in ("overall", "blur", "pose", "lighting", "size")], np.float32), q)
        assert f.is_side_face == ogates.is_side_face(f.bbox, np.float32(f.det_score))
    all_faces = app.get(frame)
    dets = np.array([list(f.bbox) + [f.det_score] for f in all_faces], np.float32)
    idx, verdict, _ = ogates.select_best(dets, [f.kps for f in all_faces])
    bf = app.best_face(frame)
    assert app.last_verdict == ("accepted", "no face", "confidence too low", "side face", "quality too low")[verdict]
    if verdict == ogates.ACCEPT:
        assert bf.det_score == max(f.det_score for f in all_faces) and np.array_equal(bf.bbox, all_faces[idx].bbox)
    else:
        assert bf is None
    from scrfd_arcface_facerecognition_amd._lib import GateConfig
    app.gate_config = GateConfig(confidence_threshold=0.0, decision_threshold=99, min_quality_threshold=0.0)     # every gate open
    assert app.best_face(frame).det_score == max(f.det_score for f in all_faces) and app.last_verdict == "accepted"
    # ... and against the fp32 oracle, not only against our own per-face API: the embedding of every returned face from the oracle's
    # warp + net on the same landmarks (detector decisions are oracle-checked on identical heads in test_gpu_models_api.py)
    from oracle import pipeline as opipe
    rec_net, rec_P = app.rec.session.net, app.rec.session.params
    for f in faces:
        ref, _ = opipe.embed(frame, f.kps, rec_net, rec_P)
        assert 1 - float(ref @ f.embedding / np.linalg.norm(ref) / np.linalg.norm(f.embedding)) < 1e-3


def test_stream_runner_equals_direct_steps():
    from scrfd_arcface_facerecognition_amd import archs
    from scrfd_arcface_facerecognition_amd._lib import Context
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet, Gallery
    from scrfd_arcface_facerecognition_amd.pipeline import FacePipeline, calibrate_detector_bias
    from scrfd_arcface_facerecognition_amd.video import StreamRunner
    from oracle import align as oalign
    ctx = Context(0)
    rng = np.random.default_rng(6)
    B, H, W = 4, 270, 480                                             # 1080p / 4: letterboxed on the device
    frames = rng.integers(0, 256, (10, H, W, 3), dtype=np.uint8)      # 10 frames = 2 full batches + a ragged tail of 2
    det_net = archs.scrfd_500m((640, 640))
    lb = np.stack([oalign.letterbox(f)[0] for f in frames[:4]])
    det_P, _ = calibrate_detector_bias(ctx, det_net, archs.synth_params(det_net, 3), lb, target=30, max_batch=4)
    rec_net = archs.mobilefacenet()
    det = CompiledNet(ctx, det_net, det_P, max_batch=B)
    rec = CompiledNet(ctx, rec_net, archs.synth_params(rec_net, 3), max_batch=B)
    gal_np = rng.standard_normal((20, 512)).astype(np.float32)
    gal = Gallery(ctx, gal_np)
    pipe = FacePipeline(ctx, det, rec, batch=B, faces_per_frame=1)
    runner = StreamRunner(pipe, gal, (H, W), similarity_thresh=0.02)
    streamed = list(runner.run(iter(frames)))
    runner.close()
    assert len(streamed) == 10
    for b0 in (0, 4, 8):
        chunk = np.zeros((B, H, W, 3), np.uint8)
        n = min(B, 10 - b0)
        chunk[:n] = frames[b0:b0 + n]
        pipe.run_step(ctx.to_device(chunk), H, W, gal, 0.02)
        direct = pipe.results(gal)
        for i in range(n):
            a, d = streamed[b0 + i], direct[i]
            assert len(a) == len(d)
            for fa, fd in zip(a, d):
                assert np.array_equal(fa[0], fd[0]) and fa[1] == fd[1] and np.array_equal(fa[2], fd[2]) and fa[3] == fd[3] and fa[4] == fd[4]
    # ... and the streamed results against the oracle (frame by frame, on the landmarks the stream reported): embedding -> similarity
    from oracle import match as omatch, pipeline as opipe
    rec_P = archs.synth_params(rec_net, 3)
    checked = 0
    for i in (0, 5, 9):                                               # one frame of each batch, the last from the ragged tail
        for bbox, score, kps, name, sim in streamed[i]:
            ref, _ = opipe.embed(frames[i], kps, rec_net, rec_P)
            j, s_ref = omatch.gallery_scan(ref, gal_np, 0.02)
            assert abs(s_ref - sim) < 2e-3
            checked += 1
    assert checked >= 1
    ctx.close()


def test_slot_uploads_overlap_compute():
    """fid_upload_async orders a copy after ALL compute enqueued so far, which serialises upload i+1 behind step i.
    The slot form (fid_upload_async_slot / _wait_slot / _release) waits only for the last reader of the SAME buffer.
    Measured here on cfg 5's per-rank share (32 frames of 1080x1920 = 199 MB per upload, SCRFD-2.5G + MobileFaceNet):
    double-buffered steps must take clearly less wall time than the same steps with the conservative ordering, and both
    orders must give identical results."""
    import ctypes as C
    import time
    from scrfd_arcface_facerecognition_amd import archs
    from scrfd_arcface_facerecognition_amd._lib import Context, check
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet, Gallery
    from scrfd_arcface_facerecognition_amd.pipeline import FacePipeline
    from scrfd_arcface_facerecognition_amd.video import _Pinned
    ctx = Context(0)
    rng = np.random.default_rng(9)
    B, H, W = 32, 1080, 1920
    det_net, rec_net = archs.ARCHS["scrfd_2.5g"]((640, 640)), archs.mobilefacenet()
    det = CompiledNet(ctx, det_net, archs.synth_params(det_net, 1), max_batch=B)
    rec = CompiledNet(ctx, rec_net, archs.synth_params(rec_net, 1), max_batch=B)
    gal = Gallery(ctx, rng.standard_normal((100, 512)).astype(np.float32))
    pipe = FacePipeline(ctx, det, rec, batch=B, faces_per_frame=1, conf_thres=0.6)
    shape = (B, H, W, 3)
    host = [_Pinned(ctx, shape, np.uint8) for _ in range(2)]
    dev = [ctx.empty(shape, np.uint8) for _ in range(2)]
    for k in range(2):
        host[k].array[:] = rng.integers(0, 256, (1, H, W, 3), dtype=np.uint8)      # one random frame broadcast: cheap to fill
        host[k].array[:, :8] = rng.integers(0, 256, (B, 8, W, 3), dtype=np.uint8)
    nb = host[0].nbytes
    steps = 8

    def serial():
        for i in range(steps):
            k = i & 1
            check(ctx.lib.fid_upload_async(ctx.handle, C.c_void_p(dev[k].ptr), C.c_void_p(host[k].ptr), nb))
            check(ctx.lib.fid_upload_wait(ctx.handle))
            pipe.run_step(dev[k], H, W, gal, 0.3)

    def overlapped():
        check(ctx.lib.fid_upload_async_slot(ctx.handle, 0, C.c_void_p(dev[0].ptr), C.c_void_p(host[0].ptr), nb))
        for i in range(steps):
            k = i & 1
            check(ctx.lib.fid_upload_wait_slot(ctx.handle, k))
            pipe.run_step(dev[k], H, W, gal, 0.3)
            check(ctx.lib.fid_upload_release(ctx.handle, k))
            if i + 1 < steps:
                check(ctx.lib.fid_upload_async_slot(ctx.handle, k ^ 1, C.c_void_p(dev[k ^ 1].ptr), C.c_void_p(host[k ^ 1].ptr), nb))

    def timed(fn):
        ctx.sync()
        t0 = time.perf_counter()
        fn()
        ctx.sync()
        return time.perf_counter() - t0

    serial(); overlapped()                                  # warm-up (autotune, first-touch of the pinned pages)
    ctx.sync()
    ts = min(timed(serial) for _ in range(3))
    def results():                                          # (detection rows past a frame's count are unspecified memory: masked out)
        cnt = pipe.post.counts.download().copy()
        det = np.where((cnt > 0)[:, None, None], pipe.post.det.download()[:, :1], 0.0)
        return cnt, pipe.idx.download().copy(), pipe.score.download().copy(), det
    res_s = results()
    to = min(timed(overlapped) for _ in range(3))
    res_o = results()
    for a, b in zip(res_s, res_o):
        assert np.array_equal(a, b)                         # the last step ran on the same buffer contents either way
    print(f"8 steps of 32 x 1080p: serial upload+compute {ts / steps * 1e3:.2f} ms/step, double-buffered {to / steps * 1e3:.2f} ms/step")
    assert to < 0.88 * ts, (to, ts)
    for h in host:
        h.free()
    ctx.close()

"""GPU: the multi-GPU step (pipeline.run_step_distributed) on a one-rank RCCL group -- exercises the real
nccl all_gather_into_tensor call, the torch-stream-backed context and the post-gather match indexing.
(The 2-rank data path is covered on CPU with gloo in test_distributed_cpu.py; N = 8 is the driver's run.)"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_one_rank_rccl_step_equals_single_gpu_step():
    import torch
    import torch.distributed as dist
    from scrfd_arcface_facerecognition_amd import archs
    from scrfd_arcface_facerecognition_amd._lib import Context
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet, Gallery
    from scrfd_arcface_facerecognition_amd.pipeline import FacePipeline, calibrate_detector_bias, run_step_distributed
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        stream = torch.cuda.Stream()
        ctx = Context(0, stream.cuda_stream)
        rng = np.random.default_rng(4)
        B, F = 4, 1
        frames = rng.integers(0, 256, (B, 320, 320, 3), dtype=np.uint8)
        det_net = archs.scrfd_500m((320, 320))
        det_P, _ = calibrate_detector_bias(ctx, det_net, archs.synth_params(det_net, 2), frames, target=32, max_batch=B)
        rec_net = archs.mobilefacenet()
        rec_P = archs.synth_params(rec_net, 2)
        det = CompiledNet(ctx, det_net, det_P, max_batch=B)
        rec = CompiledNet(ctx, rec_net, rec_P, max_batch=B * F)
        gal = Gallery(ctx, rng.standard_normal((50, 512)).astype(np.float32))
        with torch.cuda.stream(stream):
            q_local = torch.empty((B * F, 512), dtype=torch.float16, device="cuda")
            q_all = torch.empty((B * F, 512), dtype=torch.float16, device="cuda")
            pipe = FacePipeline(ctx, det, rec, batch=B, faces_per_frame=F, q_buffer=q_local)
            fd = ctx.to_device(frames)
            run_step_distributed(pipe, fd, 320, 320, gal, 0.05, q_local, q_all, dist)
            torch.cuda.synchronize()
            idx_d, sc_d = pipe.idx.download().copy(), pipe.score.download().copy()
            assert torch.equal(q_all, q_local)
            pipe.run_step(fd, 320, 320, gal, 0.05)
            torch.cuda.synchronize()
            assert np.array_equal(idx_d, pipe.idx.download()) and np.array_equal(sc_d, pipe.score.download())
            assert (pipe.post.counts.download()[:B] >= 1).all()
    finally:
        dist.destroy_process_group()


def test_native_communicator_single_rank_and_all_scopes():
    """The C-ABI's own RCCL communicator (fid_comm_unique_id / fid_comm_init_rank / fid_allgather, include/faceid.h) on a
    one-rank group, driving pipeline.run_step_distributed in its three match scopes; every scope must reproduce the
    single-GPU step.  (N = 2 runs the same product function on CPU/gloo in test_distributed_cpu.py; N = 8 is the driver's.)"""
    from scrfd_arcface_facerecognition_amd import archs
    from scrfd_arcface_facerecognition_amd._lib import Context
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet, Gallery
    from scrfd_arcface_facerecognition_amd.pipeline import (Communicator, FacePipeline, calibrate_detector_bias,
                                                            run_step_distributed)
    ctx = Context(0)
    comm = Communicator(ctx, 1, 0, lambda ident: ident)
    assert comm.get_world_size() == 1 and comm.get_rank() == 0
    rng = np.random.default_rng(4)
    B, F = 4, 2
    frames = rng.integers(0, 256, (B, 320, 320, 3), dtype=np.uint8)
    det_net = archs.scrfd_500m((320, 320))
    det_P, _ = calibrate_detector_bias(ctx, det_net, archs.synth_params(det_net, 2), frames, target=32, max_batch=B)
    rec_net = archs.mobilefacenet()
    det = CompiledNet(ctx, det_net, det_P, max_batch=B)
    rec = CompiledNet(ctx, rec_net, archs.synth_params(rec_net, 2), max_batch=B * F)
    gal_host = rng.standard_normal((333, 512)).astype(np.float32)
    gal = Gallery(ctx, gal_host)
    n = B * F
    q_local = ctx.empty((n, 512), np.float16)
    q_all = ctx.empty((n, 512), np.float16).zero()
    pipe = FacePipeline(ctx, det, rec, batch=B, faces_per_frame=F, q_buffer=q_local)
    fd = ctx.to_device(frames)
    pipe.run_step(fd, 320, 320, gal, 0.05)
    want = (pipe.idx.download().copy(), pipe.score.download().copy())
    assert (want[0] >= 0).any()
    idx_all, score_all = ctx.empty((n,), np.int32), ctx.empty((n,), np.float32)
    keys_local, keys_all = ctx.empty((n,), np.uint64), ctx.empty((1, n), np.uint64)
    for scope in ("all", "own", "sharded"):
        pipe.idx.zero(); pipe.score.zero(); idx_all.zero(); score_all.zero(); q_all.zero()
        run_step_distributed(pipe, fd, 320, 320, gal, 0.05, q_local, q_all, comm, idx_all=idx_all, score_all=score_all,
                             match_scope=scope, keys_local=keys_local, keys_all=keys_all, gallery_first_row=0, gallery_total=333)
        ctx.sync()
        assert np.array_equal(q_all.download(), q_local.download())          # the collective delivered this rank's block
        got = (pipe.idx.download(), pipe.score.download()) if scope == "own" else (idx_all.download(), score_all.download())
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), scope
    comm.close()
    ctx.close()


def test_empty_face_slots_travel_as_zero_rows():
    """SURVEY.md 8e / reference main.py:132 (only detected faces are matched): the unit-embedding matrix the all-gather moves
    carries the face counts -- slot (b, f) with f >= counts[b] is an all-zero row (fid_l2_normalize_f16_slots), never matches
    (idx -1, score 0) and gathered_face_counts recovers counts[] from the matrix alone.  Checked on synthetic counts through the
    C-ABI and on a pipeline step whose batch holds frames with fewer faces than slots."""
    import ctypes as C
    from scrfd_arcface_facerecognition_amd import archs
    from scrfd_arcface_facerecognition_amd._lib import Context, check
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet, Gallery
    from scrfd_arcface_facerecognition_amd.pipeline import FacePipeline, calibrate_detector_bias, gathered_face_counts
    ctx = Context(0)
    rng = np.random.default_rng(12)
    B, F = 5, 3
    emb = rng.standard_normal((B * F, 512)).astype(np.float32)
    counts = np.array([0, 1, 3, 2, 5], dtype=np.int32)
    gal_host = rng.standard_normal((77, 512)).astype(np.float32)
    gal_host[5] = emb[3]                                                # frame 1, slot 0: a face that matches
    gal_host[9] = emb[5]                                                # frame 1, slot 2: an EMPTY slot whose embedding would match
    gal = Gallery(ctx, gal_host)
    e_dev, c_dev = ctx.to_device(emb), ctx.to_device(counts)
    q_plain, q = ctx.empty((B * F, 512), np.float16), ctx.empty((B * F, 512), np.float16)
    check(ctx.lib.fid_l2_normalize_f16(ctx.handle, C.c_void_p(e_dev.ptr), B * F, 512, C.c_void_p(q_plain.ptr)))
    check(ctx.lib.fid_l2_normalize_f16_slots(ctx.handle, C.c_void_p(e_dev.ptr), B * F, 512, C.c_void_p(c_dev.ptr), F, C.c_void_p(q.ptr)))
    ctx.sync()
    qp, qs = q_plain.download(), q.download()
    valid = np.array([f < counts[b] for b in range(B) for f in range(F)])
    assert np.array_equal(qs[valid], qp[valid]) and (qs[~valid] == 0).all() and (np.abs(qs[valid]).max(axis=1) > 0).all()
    assert np.array_equal(gathered_face_counts(qs, B, F), np.minimum(counts, F))
    # an empty slot is {-0.0, +0.0 ...}: the same numbers as a zero row, another bit pattern than a DEGENERATE face's row (below)
    assert (qs[~valid].view(np.uint16)[:, 0] == 0x8000).all() and not qs[~valid].view(np.uint16)[:, 1:].any()
    # degenerate embeddings INSIDE a valid prefix (reference main.py:132-134 visits every detected face): an all-zero
    # embedding at slot 0 of the 3-face frame 2 and a NaN one at the last valid slot of frame 3 become all +0.0 rows -- "Unknown", but
    # COUNTED, so the faces behind them keep their slots on every rank
    emb2 = emb.copy()
    emb2[2 * F + 0] = 0
    emb2[3 * F + 1, 7] = np.nan
    gal_host2 = gal_host.copy()
    gal_host2[11] = emb[2 * F + 1]                                      # the face BEHIND the degenerate slot matches row 11
    gal2 = Gallery(ctx, gal_host2)
    e2 = ctx.to_device(emb2)
    check(ctx.lib.fid_l2_normalize_f16_slots(ctx.handle, C.c_void_p(e2.ptr), B * F, 512, C.c_void_p(c_dev.ptr), F, C.c_void_p(q.ptr)))
    ctx.sync()
    q2 = q.download()
    assert not q2[2 * F].view(np.uint16).any() and not q2[3 * F + 1].view(np.uint16).any()
    assert np.array_equal(gathered_face_counts(q2, B, F), np.minimum(counts, F))
    idx2, score2 = ctx.empty((B * F,), np.int32), ctx.empty((B * F,), np.float32)
    gal2.match_device(q, B * F, 0.05, idx2, score2)
    ctx.sync()
    i2 = idx2.download()
    assert i2[2 * F] == -1 and i2[3 * F + 1] == -1 and i2[2 * F + 1] == 11
    check(ctx.lib.fid_l2_normalize_f16_slots(ctx.handle, C.c_void_p(e_dev.ptr), B * F, 512, C.c_void_p(c_dev.ptr), F, C.c_void_p(q.ptr)))
    idx, score = ctx.empty((B * F,), np.int32), ctx.empty((B * F,), np.float32)
    gal.match_device(q, B * F, 0.05, idx, score)
    ctx.sync()
    i, s = idx.download(), score.download()
    assert (i[~valid] == -1).all() and (s[~valid] == 0).all() and i[3] == 5 and i[5] == -1

    # a pipeline step: frames 1 and 3 are featureless (constant colour); whatever the detector makes of them, slots past a frame's
    # face count must be zero rows / "Unknown", the others unit rows, and the matrix must give back the counts
    B, F = 4, 2
    frames = rng.integers(0, 256, (B, 320, 320, 3), dtype=np.uint8)
    det_net = archs.scrfd_500m((320, 320))
    det_P, _ = calibrate_detector_bias(ctx, det_net, archs.synth_params(det_net, 2), frames, target=32, max_batch=B)
    frames[1] = 0
    frames[3] = 255
    rec_net = archs.mobilefacenet()
    det = CompiledNet(ctx, det_net, det_P, max_batch=B)
    rec = CompiledNet(ctx, rec_net, archs.synth_params(rec_net, 2), max_batch=B * F)
    pipe = FacePipeline(ctx, det, rec, batch=B, faces_per_frame=F)
    pipe.run_step(ctx.to_device(frames), 320, 320, gal, 0.0)
    ctx.sync()
    cnt = np.minimum(pipe.post.counts.download()[:B], F)
    qh = pipe.q.download()
    assert np.array_equal(gathered_face_counts(qh, B, F), cnt)
    vis = np.array([f < cnt[b] for b in range(B) for f in range(F)])
    assert (pipe.idx.download()[~vis] == -1).all() and (pipe.score.download()[~vis] == 0).all()
    nrm = np.linalg.norm(qh.astype(np.float32), axis=1)
    assert np.allclose(nrm[vis], 1.0, atol=2e-3) and (nrm[~vis] == 0).all()
    res = pipe.results(gal)
    assert [len(r) for r in res] == list(cnt)
    ctx.close()

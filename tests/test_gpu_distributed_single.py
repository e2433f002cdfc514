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

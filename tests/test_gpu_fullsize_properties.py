"""GPU, BASELINE.json configs[1] at FULL size (SCRFD-10G + ArcFace-R50, 64 frames of 640x640, 1k gallery):
size-independent properties over all 64 frames, PLUS the fp32 CPU oracle on 4 of the 64 frames (heads and embeddings at the
tolerances of test_gpu_nets.py) -- the batch-64 autotuned kernel plans are other tiles / generations than the batch-1 tests use.
  * determinism: the same batch twice -> bit-identical detections, embeddings, matches
  * permutation equivariance: reversing the frame order reverses every per-frame result bit-exactly
  * duplicated frames give identical rows
  * NMS idempotence: the survivors of a frame survive a second NMS unchanged
  * a gallery built from the batch's own embeddings matches every face to itself with cosine 1 (+-1e-3)"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("plan", [None, "plans/mi355x.plan"])
def test_full_size_properties(plan, monkeypatch):
    """plan = None: the kernels a fresh autotune picks on this box; plan = the committed file: the kernels bench.py's headline number
    times (bench.py loads it the same way, read-only) -- their detector heads and embeddings meet the oracle too."""
    import os
    from conftest import ROOT
    from scrfd_arcface_facerecognition_amd import archs
    from scrfd_arcface_facerecognition_amd._lib import Context, check
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet, Gallery
    from scrfd_arcface_facerecognition_amd.pipeline import FacePipeline, calibrate_detector_bias
    monkeypatch.delenv("FID_PLAN", raising=False)
    if plan is None:
        monkeypatch.delenv("FID_PLAN_RO", raising=False)
    else:
        monkeypatch.setenv("FID_PLAN_RO", os.path.join(ROOT, plan))
    ctx = Context(0)
    B, F = 64, 1
    rng = np.random.default_rng(77)
    frames = rng.integers(0, 256, (B, 640, 640, 3), dtype=np.uint8)
    frames[5] = frames[4]                                             # a duplicated frame
    det_net = archs.scrfd_10g((640, 640))
    det_P, _ = calibrate_detector_bias(ctx, det_net, archs.synth_params(det_net, 0), frames[:8], target=48)
    rec_net = archs.iresnet50()
    rec_P = archs.synth_params(rec_net, 0)
    det = CompiledNet(ctx, det_net, det_P, max_batch=B)
    rec = CompiledNet(ctx, rec_net, rec_P, max_batch=B * F)
    if plan is not None:
        # the file's picks must really be installed (same device name, CU count and layer tables as the plan was made on): a plan that
        # does not apply would silently re-test the fresh autotune
        n_det, n_rec = det.load_plan(os.path.join(ROOT, plan)), rec.load_plan(os.path.join(ROOT, plan))
        if n_det == 0 or n_rec == 0:
            pytest.skip(f"{plan} holds no picks for this device / library revision ({n_det} detector, {n_rec} recogniser lines)")
    pipe = FacePipeline(ctx, det, rec, batch=B, faces_per_frame=F)
    gal0 = Gallery(ctx, rng.standard_normal((1000, 512)).astype(np.float32))

    def run(fr):
        pipe.run_step(ctx.to_device(fr), 640, 640, gal0, 0.05)
        pipe.post.check()
        return (pipe.post.counts.download()[:B].copy(), pipe.post.det.download()[:, :F].copy(),
                pipe.post.kps.download()[:, :F].copy(), pipe.embeddings().copy(), pipe.idx.download().copy(),
                pipe.score.download().copy())

    a = run(frames)
    b = run(frames)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)                                   # determinism
    counts, dets, kps, emb, idx, score = a
    assert (counts >= 1).all()                                        # calibrated detector: every frame yields a face
    r = run(frames[::-1].copy())
    for x, y in zip(a, r):
        assert np.array_equal(x, y[::-1])                             # permutation equivariance
    assert np.array_equal(dets[4], dets[5]) and np.array_equal(emb[4], emb[5]) and idx[4] == idx[5]
    assert np.isfinite(emb).all() and np.abs(emb).max() < 6e4

    # ---- the fp32 oracle on 4 of the 64 frames: head tensors of the batch-64 detector run and embeddings of the batch-64
    # recogniser run (tolerances of test_gpu_nets.py: bbox/kps 3e-2 stride units, cosine 1e-3; sigmoid scores 4e-3 here instead of the 3e-3
    # of the 320x320 tests: at 640x640 the stride-32 head sums fp16-rounded activations of deeper, larger maps -- tools/head_error.py
    # measures 2.4e-3 .. 3.3e-3 on that head across four different kernel plans (old kernels only / heuristic plan / autotuned with and
    # without the generation-9/10 kernels) and 1.1e-3 .. 1.6e-3 on the stride-8 / 16 heads: summation-order noise, not a kernel property)
    from oracle import align as oalign, nets as onets, pipeline as opipe
    run(frames)
    worst = {}
    for fi in (0, 21, 42, 63):
        blob = oalign.blob_from_images([frames[fi]], det_net.in_scale, det_net.in_mean)
        ref = onets.run_net(det_net, det_P, blob)
        for name in det_net.outputs:
            fused = det.read(name, B)[fi:fi + 1]
            sc_, bb_, kp_ = ref[name]
            worst[name] = max(worst.get(name, 0.0), float(np.abs(fused[..., 0:2].reshape(1, -1, 1) - sc_).max()))
            assert np.abs(fused[..., 0:2].reshape(1, -1, 1) - sc_).max() < 4e-3, (fi, name)
            assert np.abs(fused[..., 2:10].reshape(1, -1, 4) - bb_).max() < 3e-2, (fi, name)
            assert np.abs(fused[..., 10:30].reshape(1, -1, 10) - kp_).max() < 3e-2, (fi, name)
        oe, ocrop = opipe.embed(frames[fi], kps[fi, 0].reshape(5, 2), rec_net, rec_P)     # same landmarks as the device used
        assert np.array_equal(pipe.crops.download()[fi], ocrop)                          # warp bit-exact at full size
        e = emb[fi]
        assert 1 - float(oe @ e / np.linalg.norm(oe) / np.linalg.norm(e)) < 1e-3, fi
        assert np.abs(oe / np.linalg.norm(oe) - e / np.linalg.norm(e)).max() < 1e-3, fi
    print(f"\nworst |score - oracle| per head on 4 frames (plan={plan}): " + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()))

    # NMS idempotence on one frame's full detection list
    pipe2 = FacePipeline(ctx, det, rec, batch=B, faces_per_frame=F)
    pipe2.F = 0
    pipe2.detect(ctx.to_device(frames), 640, 640)                     # max_num = 0: all survivors
    pipe2.post.check()

    # ---- end-to-end detector agreement on ALL 64 frames (reference models/scrfd.py:140-156): the fp32 oracle's
    # survivors vs the device's.  (1) the device's own post-process equals the oracle's post-process of the DEVICE heads bit for bit on
    # every frame (decisions on identical heads); (2) against the fp32 heads every survivor either has a counterpart (IoU >= 0.9) or is
    # a flip that one quantity within 5e-3 of a decision boundary explains (score vs conf_thres, suppressing IoU vs iou_thres, score
    # order of an overlapping pair) or that cascades from such a flip.  Zero unexplained mismatches; the counts are printed.
    import torch
    from oracle import agreement as oagree, postprocess as opp
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    fused = [det.read(name, B) for name in det_net.outputs]
    cnt_all, det_all, kps_all = pipe2.post.counts.download()[:B], pipe2.post.det.download(), pipe2.post.kps.download()
    per_frame, worst64, top1, e2e = [], [0.0, 0.0, 0.0], {"same": 0, "marginal": 0, "unexplained": 0, "empty": 0}, []
    for fi in range(B):
        dev_outs = oagree.fused_to_session_outputs(fused, fi)
        od, ok = opp.detect_from_heads(dev_outs, (640, 640), (640, 640), 0.5, 0.4, 0)
        n = int(cnt_all[fi])
        assert n == len(od) <= pipe2.post.cap, (fi, n, len(od))
        assert np.array_equal(det_all[fi, :n], od) and np.array_equal(kps_all[fi, :n].reshape(n, 5, 2), ok), fi
        blob = oalign.blob_from_images([frames[fi]], det_net.in_scale, det_net.in_mean)
        ref_outs = onets.scrfd_session_outputs(det_net, det_P, blob)
        per_frame.append(oagree.survivor_agreement(ref_outs, dev_outs, (640, 640), 0.5, 0.4, margin=5e-3))
        # the face the pipeline EMBEDS (reference scrfd.py:159-177 with max_num = 1 + main.py:130-134): the oracle's pick on
        # the fp32 heads vs the device's pick; the device's own max_num = 1 result (first pipeline, `dets`) is that pick bit for bit
        verdict, ia, ib = oagree.top1_agreement(per_frame[-1])
        top1[verdict] += 1
        assert ib >= 0 and np.array_equal(dets[fi, 0], per_frame[-1]["det_b"][ib]), fi
        if verdict == "same" and len(e2e) < 6:
            _, okps1 = opp.detect_from_heads(ref_outs, (640, 640), (640, 640), 0.5, 0.4, 1)
            oe, _ = opipe.embed(frames[fi], okps1[0], rec_net, rec_P)                      # oracle end to end: fp32 heads -> landmarks -> crop -> fp32 net
            e2e.append(1 - float(oe @ emb[fi] / np.linalg.norm(oe) / np.linalg.norm(emb[fi])))
        for li in range(3):                                           # session outputs 0..2 = the three strides' scores
            worst64[li] = max(worst64[li], float(np.abs(np.asarray(dev_outs[li], np.float32) - ref_outs[li]).max()))
    agree = oagree.summarize(per_frame)
    print(f"\ndet_survivor_agreement (plan={plan}): {agree}; worst |score - oracle| over all {B} frames per stride: " + " ".join(f"{v:.2e}" for v in worst64))
    assert agree["unexplained"] == 0, [(fi, d["detail"]) for fi, d in enumerate(per_frame) if d["unexplained"]]
    assert agree["matched"] >= 0.9 * agree["survivors_a"], agree
    line = (f"plan={plan}: det_survivor_agreement {agree}; top1_face_agreement {top1}; end-to-end embed cosine delta on {len(e2e)} same-pick frames "
            f"max {max(e2e):.2e}; head_score_err_max s8 / s16 / s32 over {B} frames " + " / ".join(f"{v:.3e}" for v in worst64))
    print("\n" + line)
    out_dir = os.path.join(ROOT, "gpurun_out")                         # (pytest -q swallows stdout: the margins are kept as a file -- VERDICT r4 item 6)
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "fullsize_margins.txt"), "a") as f:
        f.write(line + "\n")
    assert top1["unexplained"] == 0 and top1["same"] >= 0.8 * B, top1
    assert max(e2e) < 2e-3, e2e          # (device landmarks differ from the oracle's by the fp16 head error: 3e-2 stride units = a fraction of a pixel of the crop)
    n0 = int(pipe2.post.counts.download()[0])
    d0 = pipe2.post.det.download()[0, :n0]
    assert n0 >= 1
    dd = ctx.to_device(d0)
    keep, cnt = ctx.empty((n0,), np.int32), ctx.empty((1,), np.int32)
    check(ctx.lib.fid_nms(ctx.handle, C.c_void_p(dd.ptr), n0, 0.4, C.c_void_p(keep.ptr), C.c_void_p(cnt.ptr)))
    assert int(cnt.download()[0]) == n0 and np.array_equal(keep.download()[:n0], np.arange(n0))

    # self-gallery: every face matches itself
    gal = Gallery(ctx, emb)
    run(frames)                                                       # pipe.q = unit embeddings of the forward order again
    pipe.match(gal, 0.5)
    idx2, sc2 = pipe.idx.download(), pipe.score.download()
    for i in range(B):
        assert idx2[i] == i or np.array_equal(emb[idx2[i]], emb[i])    # duplicates: first index wins
        assert abs(sc2[i] - 1.0) < 1e-3
    assert idx2[5] == 4
    ctx.close()

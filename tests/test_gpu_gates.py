"""GPU parity: fid_face_gates (csrc/gates.hip) -- quality scores, side-face test and best-face verdict of the reference's product
layer (smart_face_recognition.py:1145-1216,1218-1297,1299-1399,1473-1519) -- against the reference-generated golden vectors
(tests/golden/gates.npz, bit for bit) and against oracle/gates.py on ragged batches."""
import json

import numpy as np
import pytest

from conftest import load_golden
from oracle import gates as ogates

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from scrfd_arcface_facerecognition_amd._lib import Context
    c = Context(0)
    yield c
    c.close()


def run(ctx, det, kps, counts, F, pose=None, config=None):
    from scrfd_arcface_facerecognition_amd.app import face_gates
    B, cap = det.shape[:2]
    return face_gates(ctx, ctx.to_device(det), ctx.to_device(kps.reshape(B, cap, 10)), ctx.to_device(counts.astype(np.int32)), B, cap, F,
                      config, pose)


def test_gates_equal_the_reference_vectors(ctx):
    """the 600 golden faces as 40 frames of 15 slots: quality bit for bit, side score and flag exact -- with and without pose angles"""
    from scrfd_arcface_facerecognition_amd._lib import GateConfig
    g = load_golden("gates.npz")
    cfg = GateConfig.from_reference_json(json.loads(bytes(g["config"]).decode()))
    B, F = 40, 15
    det = np.concatenate([g["bbox"], g["score"][:, None]], 1).reshape(B, F, 5).astype(np.float32)
    kps = g["kps"].reshape(B, F, 5, 2)
    counts = np.full(B, F)
    q, score, flag, best = run(ctx, det, kps, counts, F, pose=g["pose"].reshape(B, F, 2), config=cfg)
    assert np.array_equal(q.reshape(-1, 5).astype(np.float64), g["quality"])
    assert np.array_equal(score.reshape(-1), g["bbox_side"][:, 1])
    assert np.array_equal(flag.reshape(-1).astype(np.int32), g["side"])                     # is_side_face incl. the pose-angle branch
    q2, score2, flag2, _ = run(ctx, det, kps, counts, F, config=cfg)                       # no pose data: the bbox analysis decides everywhere
    assert np.array_equal(q2, q) and np.array_equal(score2, score)
    assert np.array_equal(flag2.reshape(-1).astype(np.int32), g["bbox_side"][:, 0])
    for b in range(B):                                                                       # the best face and its verdict per frame
        idx, verdict, _ = ogates.select_best(det[b], kps[b], poses=g["pose"].reshape(B, F, 2)[b])
        assert (int(best[b, 0]), int(best[b, 1])) == (idx, verdict), b


def test_pose_angles_at_the_threshold_in_float64(ctx):
    """the reference hands python floats (float64 radians) to math.degrees (smart_face_recognition.py:1226-1240); the angles cross
    the C-ABI as float64 too, so an angle a few ulps either side of the 35-degree threshold gets the reference's verdict -- in float32
    radians(35) rounds to 34.9999998 and radians(36) to 36.000001 degrees, i.e. the verdict of an angle AT the threshold would depend on
    which way float32 rounds it."""
    import math
    cfg = ogates.DEFAULT_CONFIG
    thr = float(cfg["yaw_threshold"])
    base = math.radians(thr)
    angles, up, dn = [base], base, base
    for _ in range(39):                                      # +-1 .. +-39 ulps of float64 around the threshold (float32 cannot tell them apart)
        up, dn = float(np.nextafter(up, 1.0)), float(np.nextafter(dn, 0.0))
        angles += [up, dn]
    angles += [-a for a in angles[:9]] + [float(np.float32(base)), float(np.nextafter(np.float32(base), np.float32(1)))]
    n = len(angles)
    det = np.tile(np.array([100, 100, 220, 260, 0.9], np.float32), (1, n, 1))
    kps = np.tile(np.array([[130, 150], [190, 150], [160, 190], [140, 220], [180, 220]], np.float32), (1, n, 1, 1))
    for col in (0, 1):                                       # yaw, then pitch
        pose = np.zeros((1, n, 2), np.float64)
        pose[0, :, col] = angles
        _, _, flag, _ = run(ctx, det, kps, np.array([n]), n, pose=pose)
        want = [ogates.is_side_face(det[0, i, :4], 0.9, *(pose[0, i]), cfg) for i in range(n)]
        assert flag.reshape(-1).tolist() == [bool(w) for w in want], col
        assert any(want) and not all(want)                   # the sweep really straddles the threshold


@pytest.mark.parametrize("seed", range(3))
def test_gates_on_ragged_batches(ctx, seed):
    """counts of 0 .. cap faces per frame, more detections than face slots (cap > F), ties of the top score, custom thresholds"""
    from scrfd_arcface_facerecognition_amd._lib import GateConfig
    rng = np.random.default_rng(70 + seed)
    B, cap, F = 9, 12, 8
    xy = rng.uniform(0, 900, (B, cap, 2)); wh = rng.uniform(10, 500, (B, cap, 2))
    det = np.concatenate([xy, xy + wh, rng.choice([0.3, 0.55, 0.6, 0.75, 0.9], (B, cap, 1))], 2).astype(np.float32)
    kps = (xy[:, :, None, :] + rng.uniform(0, 1, (B, cap, 5, 2)) * wh[:, :, None, :]).astype(np.float32)
    counts = np.array([0, 1, 2, 5, 8, 9, 12, 3, 7])
    kw = dict(confidence_threshold=0.5 + 0.1 * seed, decision_threshold=3 + seed, min_quality_threshold=0.3)
    q, score, flag, best = run(ctx, det, kps, counts, F, config=GateConfig(**kw))
    cfg = dict(ogates.DEFAULT_CONFIG, **kw)
    for b in range(B):
        n = min(counts[b], F)
        for f in range(F):
            if f < n:
                assert np.array_equal(q[b, f], ogates.face_quality(det[b, f, :4], kps[b, f], det[b, f, 4], cfg))
                d = det[b, f]
                fl, sc = ogates.bbox_side_score(d[2] - d[0], d[3] - d[1], d[1], d[0], d[4], cfg)
                assert (bool(flag[b, f]), int(score[b, f])) == (fl, sc)
            else:
                assert not q[b, f].any() and score[b, f] == 0 and not flag[b, f]           # empty slots: zeros
        idx, verdict, _ = ogates.select_best(det[b, :n], kps[b, :n], cfg)
        assert (int(best[b, 0]), int(best[b, 1])) == (idx, verdict), (b, n)

"""Oracle-side checker: how far do TWO sets of SCRFD head tensors (the fp32 oracle's and the device's fp16 net's) agree in the
decisions reference models/scrfd.py:140-156 takes on them (threshold, sort, NMS)?  Test infrastructure only (tests/, bench.py's
checker legs): nothing here is on the product path.

fp16 heads cannot reproduce fp32 decisions for candidates that sit within the head error of a decision boundary, so
the comparison is a classification, not an equality: every survivor of one side that has no counterpart (IoU >= 0.9) on the other
side is either

  marginal   one quantity within `margin` of a boundary explains it: its score vs conf_thres (scrfd.py:109), the IoU of the pair that
             suppresses it vs iou_thres (scrfd.py:204), or the score ORDER of two overlapping candidates (scrfd.py:144,188: whichever
             ranks first suppresses the other),
  cascade    it is suppressed by (or owes its survival to the absence of) a survivor that is itself a marginal flip, or
  unexplained  none of the above -- a real disagreement.  The tests assert there is none.
"""
import numpy as np

from . import postprocess as pp


def fused_to_session_outputs(fused_per_level, frame):
    """the device's fused fp32 head tensors [B, H, W, 32] (cls 2 | bbox 8 | kps 20; csrc/net.h) of ONE frame in the 9-array order of
    SCRFD's session.run (scores x3, bbox x3, kps x3; scrfd.py:83-94)"""
    sc = [f[frame, ..., 0:2].reshape(-1, 1) for f in fused_per_level]
    bb = [f[frame, ..., 2:10].reshape(-1, 4) for f in fused_per_level]
    kp = [f[frame, ..., 10:30].reshape(-1, 10) for f in fused_per_level]
    return sc + bb + kp


def _iou_matrix(a, b):
    """IoU with the reference's +1 pixel convention (scrfd.py:186,199-204), [len(a), len(b)] float64"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if len(a) == 0 or len(b) == 0:
        return np.zeros((len(a), len(b)))
    area_a = (a[:, 2] - a[:, 0] + 1) * (a[:, 3] - a[:, 1] + 1)
    area_b = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
    w = np.maximum(0.0, np.minimum(a[:, None, 2], b[None, :, 2]) - np.maximum(a[:, None, 0], b[None, :, 0]) + 1)
    h = np.maximum(0.0, np.minimum(a[:, None, 3], b[None, :, 3]) - np.maximum(a[:, None, 1], b[None, :, 1]) + 1)
    inter = w * h
    return inter / (area_a[:, None] + area_b[None, :] - inter)


def _candidates(outs, input_size, conf):
    """all candidates with score >= conf of one frame: boxes [K,4] (input-image pixels), scores [K]"""
    s, b, _ = pp.decode_heads(outs, (input_size[1], input_size[0]), conf)
    return np.vstack(b).astype(np.float64), np.vstack(s).ravel().astype(np.float64)


def survivor_agreement(outs_a, outs_b, input_size=(640, 640), conf_thres=0.5, iou_thres=0.4, margin=5e-3, match_iou=0.9):
    """One frame, two sets of session outputs (a = reference side, b = device side).  Returns a dict of counts:
    survivors_a, survivors_b, matched, marginal_flips, cascade_flips, unexplained (+ `detail`: one tuple per unmatched survivor)."""
    det_a, _ = pp.detect_from_heads(outs_a, (input_size[1], input_size[0]), input_size, conf_thres, iou_thres, 0)
    det_b, _ = pp.detect_from_heads(outs_b, (input_size[1], input_size[0]), input_size, conf_thres, iou_thres, 0)
    m = _iou_matrix(det_a[:, :4], det_b[:, :4])
    pair_a, pair_b = -np.ones(len(det_a), int), -np.ones(len(det_b), int)
    if m.size:
        for flat in np.argsort(-m, axis=None):                        # greedy one-to-one matching, best IoU first
            i, j = divmod(int(flat), m.shape[1])
            if m[i, j] < match_iou:
                break
            if pair_a[i] < 0 and pair_b[j] < 0:
                pair_a[i], pair_b[j] = j, i
    sides = {"a": (det_a, pair_a, outs_a, det_b, pair_b, outs_b), "b": (det_b, pair_b, outs_b, det_a, pair_a, outs_a)}
    cands = {k: _candidates(v[2], input_size, conf_thres - margin) for k, v in sides.items()}
    verdict = {}                                                       # (side, index) -> "marginal:<why>" | "cascade" | None (pending)
    pending = []
    for side, (det, pair, _, odet, opair, _) in sides.items():
        other = "b" if side == "a" else "a"
        ocb, ocs = cands[other]                                        # the OTHER side's candidates (down to conf - margin)
        for i in np.where(pair < 0)[0]:
            x, sx = det[i, :4], float(det[i, 4])
            if abs(sx - conf_thres) < margin:
                verdict[(side, i)] = "marginal:score"; continue
            io = _iou_matrix(x[None], ocb)[0] if len(ocb) else np.zeros(0)
            if len(io) == 0 or io.max() < match_iou:                   # no counterpart candidate over there and the score is not marginal
                verdict[(side, i)] = None; pending.append((side, i, None, None)); continue
            k = int(io.argmax())
            xo, so = ocb[k], float(ocs[k])                             # X as the other side sees it
            if so < conf_thres + margin:
                verdict[(side, i)] = "marginal:score"; continue
            # over there X' exists, clears the threshold and still did not survive: something that survived suppresses it
            ios = _iou_matrix(xo[None], odet[:, :4])[0] if len(odet) else np.zeros(0)
            sup = [j for j in range(len(odet)) if ios[j] > iou_thres - margin]
            why = None
            for j in sup:
                if abs(ios[j] - iou_thres) < margin:
                    why = "marginal:iou"; break
                j_here = opair[j]                                      # the suppressor's counterpart on X's own side
                if j_here >= 0:
                    iou_here = _iou_matrix(x[None], det[j_here:j_here + 1, :4])[0, 0]
                    if abs(iou_here - iou_thres) < margin:
                        why = "marginal:iou"; break
                    if abs(float(odet[j, 4]) - so) < margin or abs(float(det[j_here, 4]) - sx) < margin:
                        why = "marginal:rank"; break
                elif abs(float(odet[j, 4]) - so) < margin:
                    why = "marginal:rank"; break
            if why:
                verdict[(side, i)] = why; continue
            verdict[(side, i)] = None
            pending.append((side, i, xo, [j for j in sup if opair[j] < 0]))   # suppressors that are themselves flips of the other side
    # cascades: X is explained when one of its suppressors over there is an (explained) flip, or -- with no counterpart at all -- when
    # it overlaps an explained flip of the other side (the thing that would have suppressed it here is missing here)
    changed = True
    while changed:
        changed = False
        for side, i, xo, flips in pending:
            if verdict[(side, i)] is not None:
                continue
            other = "b" if side == "a" else "a"
            odet = sides[side][3]
            if flips is None:
                ios = _iou_matrix(sides[side][0][i:i + 1, :4], odet[:, :4])[0] if len(odet) else np.zeros(0)
                flips = [j for j in range(len(odet)) if ios[j] > iou_thres - margin and sides[side][4][j] < 0]
            if any(verdict.get((other, j)) is not None for j in flips):
                verdict[(side, i)] = "cascade"; changed = True
    detail = [(s, int(i), v or "unexplained") for (s, i), v in sorted(verdict.items())]
    return {"det_a": det_a, "det_b": det_b, "pair_a": pair_a, "pair_b": pair_b,
            "survivors_a": int(len(det_a)), "survivors_b": int(len(det_b)), "matched": int((pair_a >= 0).sum()),
            "marginal_flips": sum(1 for _, _, v in detail if v.startswith("marginal")),
            "cascade_flips": sum(1 for _, _, v in detail if v == "cascade"),
            "unexplained": sum(1 for _, _, v in detail if v == "unexplained"), "detail": detail}


def summarize(per_frame):
    """sum of the per-frame dicts of survivor_agreement (without `detail`)"""
    keys = ("survivors_a", "survivors_b", "matched", "marginal_flips", "cascade_flips", "unexplained")
    out = {k: int(sum(d[k] for d in per_frame)) for k in keys}
    out["frames"] = len(per_frame)
    return out


def _area(d):
    return (d[:, 2] - d[:, 0]) * (d[:, 3] - d[:, 1])                  # scrfd.py:160 (no +1 here)


def top1_agreement(agree, area_tol=1e-2, match_iou=0.9):
    """Which face does the pipeline EMBED?  reference scrfd.py:159-177 with max_num = 1, metric "max": of the NMS survivors the one of largest
    box AREA (main.py:130-134 then aligns and embeds exactly that face).  `agree` = survivor_agreement's result for the frame (a = fp32
    oracle heads, b = device heads).  Verdicts:

      same         both sides pick counterparts of each other (IoU >= match_iou)
      marginal     the picks differ and either (1) on one side the two rivals' areas are within `area_tol` (relative): a box error of the size
                   the fp16 heads carry (3e-2 stride units on a box of tens of pixels) reorders them, or (2) one side's pick has no counterpart
                   on the other side and survivor_agreement explains that flip (marginal / cascade)
      unexplained  anything else -- the tests assert there is none
      empty        neither side has a survivor (no face to embed); one side only: judged like (2)

    Returns (verdict, index of a's pick or -1, index of b's pick or -1)."""
    det_a, det_b, pair_a, pair_b = agree["det_a"], agree["det_b"], agree["pair_a"], agree["pair_b"]
    why = {(s_, i): v for s_, i, v in agree["detail"]}
    if len(det_a) == 0 and len(det_b) == 0:
        return "empty", -1, -1
    ia = int(np.argsort(_area(det_a), kind="stable")[::-1][0]) if len(det_a) else -1
    ib = int(np.argsort(_area(det_b), kind="stable")[::-1][0]) if len(det_b) else -1
    if ia >= 0 and ib >= 0 and pair_a[ia] == ib:
        return "same", ia, ib

    def explained(side, i):
        return i < 0 or why.get((side, i), "unexplained") != "unexplained"
    if ia < 0 or ib < 0:
        return ("marginal" if explained("a", ia) and explained("b", ib) and (ia < 0 or pair_a[ia] < 0) and (ib < 0 or pair_b[ib] < 0) else "unexplained"), ia, ib
    ja, jb = int(pair_a[ia]), int(pair_b[ib])                          # a's pick as b sees it, b's pick as a sees it
    if ja < 0 or jb < 0:                                              # a pick that the other side does not have at all
        ok = (ja >= 0 or explained("a", ia)) and (jb >= 0 or explained("b", ib))
        return ("marginal" if ok else "unexplained"), ia, ib
    ar_a, ar_b = _area(det_a), _area(det_b)
    close_a = abs(float(ar_a[ia]) - float(ar_a[jb])) <= area_tol * float(ar_a[ia])      # on side a: its pick vs b's pick
    close_b = abs(float(ar_b[ib]) - float(ar_b[ja])) <= area_tol * float(ar_b[ib])
    return ("marginal" if close_a or close_b else "unexplained"), ia, ib

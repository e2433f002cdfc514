"""Oracle: the product layer's face gates.  Follows reference
smart_face_recognition.py:1145-1216 (assess_face_quality), :1218-1297 (get_face_pose_angles / is_side_face),
:1299-1399 (analyze_bbox_for_side_face) and :1473-1519 (best face + the four rejections), with the thresholds of the
reference's config.json (blocks face_detection, face_quality, side_face_detection; :130 for the confidence threshold).
Test infrastructure only.

Arithmetic: the face fields are float32 (insightface's bbox / kps / det_score) and every constant of the reference is a
python number, so under NumPy >= 2 each operation runs in float32 with the constant rounded to float32 first; `min(1.0, x)`
keeps x when x < 1.  tests/golden/gates.npz (tools/gen_golden_gates.py: the reference's own methods, NumPy 2.2) pins this
restatement bit for bit.  The pose angles (radians, optional) stay python floats: degrees in float64."""
import math

import numpy as np

F = np.float32

# reference config.json (face_detection / face_quality / side_face_detection), flattened; the order is the C struct's (include/faceid.h)
DEFAULT_CONFIG = {
    "size_normalization": 10000.0,
    "w_detection": 0.4, "w_size": 0.2, "w_blur": 0.2, "w_pose": 0.1, "w_lighting": 0.1,
    "ar_extreme_profile": 0.2, "ar_very_strong_profile": 0.3, "ar_strong_profile": 0.5,
    "ar_very_wide": 2.5, "ar_wide": 2.0, "ar_moderately_wide": 1.6,
    "area_extremely_small": 1200.0, "area_very_small": 1800.0, "area_small": 2500.0, "area_very_large": 400000.0, "area_large": 300000.0,
    "compactness_very_low": 0.10, "compactness_low": 0.6,
    "confidence_very_low": 0.15, "confidence_low": 0.7,
    "edge_position_threshold": 30.0, "decision_threshold": 4,
    "yaw_threshold": 35.0, "pitch_threshold": 35.0,
    "confidence_threshold": 0.6, "min_quality_threshold": 0.05,
}


def config_from_reference_json(cfg):
    """the three blocks of the reference's config.json -> the flat dict above"""
    q, s, d = cfg["face_quality"], cfg["side_face_detection"], cfg["face_detection"]
    a, ar, c, cf = s["aspect_ratio_thresholds"], s["area_thresholds"], s["compactness_thresholds"], s["confidence_thresholds"]
    return {
        "size_normalization": float(q["size_normalization"]),
        "w_detection": q["weights"]["detection_score"], "w_size": q["weights"]["size_score"], "w_blur": q["weights"]["blur_score"],
        "w_pose": q["weights"]["pose_score"], "w_lighting": q["weights"]["lighting_score"],
        "ar_extreme_profile": a["extreme_profile"], "ar_very_strong_profile": a["very_strong_profile"], "ar_strong_profile": a["strong_profile"],
        "ar_very_wide": a["very_wide"], "ar_wide": a["wide"], "ar_moderately_wide": a["moderately_wide"],
        "area_extremely_small": float(ar["extremely_small"]), "area_very_small": float(ar["very_small"]), "area_small": float(ar["small"]),
        "area_very_large": float(ar["very_large"]), "area_large": float(ar["large"]),
        "compactness_very_low": c["very_low"], "compactness_low": c["low"],
        "confidence_very_low": cf["very_low"], "confidence_low": cf["low"],
        "edge_position_threshold": float(s["edge_position_threshold"]), "decision_threshold": int(s["decision_threshold"]),
        "yaw_threshold": d["yaw_threshold"], "pitch_threshold": d["pitch_threshold"],
        "confidence_threshold": d["confidence_threshold"], "min_quality_threshold": d["min_quality_threshold"],
    }


def _cap1(x):
    """min(1.0, x) of the reference: x stays when x < 1 (NaN: not < 1, so 1.0)"""
    return x if x < F(1.0) else F(1.0)


def face_quality(bbox, kps, det_score, cfg=DEFAULT_CONFIG):
    """:1145-1216 -> (overall, blur, pose, lighting, size) as float32"""
    bbox = np.asarray(bbox, F); det = F(det_score)
    area = (bbox[2] - bbox[0]) * (bbox[3] - bbox[1])
    size = _cap1(area / F(cfg["size_normalization"]))
    blur = _cap1(det * F(1.2))
    pose = F(1.0)
    if kps is not None and len(kps) >= 5:
        k = np.asarray(kps, F)
        pose = _cap1(((k[:, 0].max() - k[:, 0].min()) + (k[:, 1].max() - k[:, 1].min())) / F(100))
    light = _cap1(det * F(1.1))
    overall = det * F(cfg["w_detection"]) + size * F(cfg["w_size"])
    overall = overall + blur * F(cfg["w_blur"])
    overall = overall + pose * F(cfg["w_pose"])
    overall = overall + light * F(cfg["w_lighting"])
    return np.array([overall, blur, pose, light, size], F)


def bbox_side_score(width, height, top, left, det_score, cfg=DEFAULT_CONFIG):
    """:1299-1399 -> (is_side_face, score)"""
    w, h, top, left, det = F(width), F(height), F(top), F(left), F(det_score)
    if w <= 0 or h <= 0:
        return False, 0
    ratio, area, perim = w / h, w * h, F(2) * (w + h)
    comp = (F(4 * 3.14159) * area) / (perim * perim) if perim > 0 else F(0)
    s = 0
    if ratio < F(cfg["ar_extreme_profile"]): s += 4
    elif ratio < F(cfg["ar_very_strong_profile"]): s += 3
    elif ratio < F(cfg["ar_strong_profile"]): s += 2
    elif ratio > F(cfg["ar_very_wide"]): s += 3
    elif ratio > F(cfg["ar_wide"]): s += 2
    elif ratio > F(cfg["ar_moderately_wide"]): s += 1
    if area < F(cfg["area_extremely_small"]): s += 3
    elif area < F(cfg["area_very_small"]): s += 2
    elif area < F(cfg["area_small"]): s += 1
    elif area > F(cfg["area_very_large"]): s += 2
    elif area > F(cfg["area_large"]): s += 1
    if comp < F(cfg["compactness_very_low"]): s += 2
    elif comp < F(cfg["compactness_low"]): s += 1
    if det != 0 and det < F(cfg["confidence_very_low"]): s += 2
    elif det != 0 and det < F(cfg["confidence_low"]): s += 1
    if left < F(cfg["edge_position_threshold"]) or top < F(cfg["edge_position_threshold"]): s += 1
    return s >= int(cfg["decision_threshold"]), s


def is_side_face(bbox, det_score, yaw=0.0, pitch=0.0, cfg=DEFAULT_CONFIG):
    """:1248-1297: the pose angles (radians; 0 = not available) decide when either is non-zero, else the bbox analysis"""
    yaw_deg = abs(math.degrees(yaw)) if yaw else 0.0
    pitch_deg = abs(math.degrees(pitch)) if pitch else 0.0
    if yaw_deg > 0 or pitch_deg > 0:
        return yaw_deg > cfg["yaw_threshold"] or pitch_deg > cfg["pitch_threshold"]
    b = np.asarray(bbox, F)
    return bbox_side_score(b[2] - b[0], b[3] - b[1], b[1], b[0], det_score, cfg)[0]


# verdicts of select_best (the reference returns None with a log line for 1..4)
ACCEPT, NO_FACE, LOW_CONFIDENCE, SIDE_FACE, LOW_QUALITY = 0, 1, 2, 3, 4


def select_best(dets, kpss, cfg=DEFAULT_CONFIG, poses=None):
    """:1473-1519 on one image's faces (dets [K, 5] = bbox + score, kpss [K, 5, 2]): the FIRST face with the highest det_score,
    then confidence / side-face / quality rejections in the reference's order.  -> (index or -1, verdict, quality [5] or None)"""
    if len(dets) == 0:
        return -1, NO_FACE, None
    best = 0
    for i in range(1, len(dets)):               # python's max(): the first maximal element
        if dets[i][4] > dets[best][4]:
            best = i
    d = np.asarray(dets[best], F)
    if d[4] < F(cfg["confidence_threshold"]):
        return best, LOW_CONFIDENCE, None
    yaw, pitch = (poses[best] if poses is not None else (0.0, 0.0))
    if is_side_face(d[:4], d[4], yaw, pitch, cfg):
        return best, SIDE_FACE, None
    q = face_quality(d[:4], kpss[best], d[4], cfg)
    if q[0] < F(cfg["min_quality_threshold"]):
        return best, LOW_QUALITY, q
    return best, ACCEPT, q

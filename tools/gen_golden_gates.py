#!/usr/bin/env python3
"""Golden vectors for the product layer's face gates from the reference's own code.

Runs ONLY in the build container (needs --reference /root/reference).  `smart_face_recognition.py` imports cv2, insightface and
qdrant_client (absent here) and builds its FastAPI app at import time, so inert stub modules are registered first (as tools/gen_golden.py does): they hold NO
arithmetic and none of their attributes is touched by the functions under test.  The three methods are pure functions of the
face's bbox / kps / det_score (/ yaw / pitch) and of the `face_quality`, `side_face_detection`, `face_detection` blocks of the
reference's config.json; they are called unbound on a plain namespace that carries that config and a logger:

  SmartFaceRecognition.assess_face_quality          smart_face_recognition.py:1145-1216
  SmartFaceRecognition.is_side_face                 smart_face_recognition.py:1248-1297  (-> get_face_pose_angles :1218-1246)
  SmartFaceRecognition.analyze_bbox_for_side_face   smart_face_recognition.py:1299-1399

Face fields are numpy float32 like insightface's (bbox [4], kps [5, 2], det_score), yaw / pitch python floats in radians.
Output: tests/golden/gates.npz (inputs + expected outputs + the config values used; data only).
"""
import argparse
import json
import logging
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True


class _Anything:
    """inert placeholder: constructible with any arguments, every attribute is another placeholder, usable as a decorator factory
    (`@app.get("/")` hands the function back unchanged)"""
    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything()

    def __call__(self, *a, **k):
        return a[0] if len(a) == 1 and not k and callable(a[0]) and not isinstance(a[0], _Anything) else self


class _Inert(types.ModuleType):
    """module whose every attribute is the inert placeholder class"""
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything


def install_stubs():
    # absent third-party packages (cv2, insightface, qdrant_client) and the web front end (the module builds its FastAPI app at import time
    # against ./static and ./templates of its own directory): none of them is touched by the three methods under test
    for name in ("cv2", "insightface", "insightface.app", "qdrant_client", "qdrant_client.http", "qdrant_client.http.models",
                 "fastapi", "fastapi.responses", "fastapi.staticfiles", "fastapi.templating", "uvicorn"):
        sys.modules[name] = _Inert(name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "gates.npz"))
    args = ap.parse_args()
    install_stubs()
    sys.path.insert(0, args.reference)
    import smart_face_recognition as S
    with open(os.path.join(args.reference, "config.json")) as f:
        cfg = json.load(f)
    me = types.SimpleNamespace(config=cfg, logger=logging.getLogger("gates"))
    K = S.SmartFaceRecognition
    for m in ("assess_face_quality", "is_side_face", "get_face_pose_angles", "analyze_bbox_for_side_face"):
        setattr(me, m, types.MethodType(getattr(K, m), me))

    rng = np.random.default_rng(20251004)
    n = 600
    bbox = np.zeros((n, 4), np.float32)
    kps = np.zeros((n, 5, 2), np.float32)
    score = np.zeros(n, np.float32)
    pose = np.zeros((n, 2), np.float64)          # yaw, pitch in radians; 0 = not available (the reference's own convention)
    for i in range(n):
        kind = i % 6
        w = float(rng.uniform(8, 700)); h = float(rng.uniform(8, 700))
        if kind == 1:                              # profiles: narrow / wide boxes around the aspect thresholds
            h = float(rng.uniform(40, 400)); w = h * float(rng.choice([0.15, 0.2, 0.25, 0.3, 0.45, 0.5, 0.55, 1.6, 1.7, 2.0, 2.2, 2.5, 2.8]))
        elif kind == 2:                            # areas around the area thresholds
            a = float(rng.choice([1000, 1200, 1500, 1800, 2200, 2500, 3000, 290000, 300000, 350000, 400000, 450000]))
            r = float(rng.uniform(0.7, 1.4)); w = (a * r) ** 0.5; h = a / w
        x1 = float(rng.uniform(0, 60)) if kind == 3 else float(rng.uniform(0, 1200))
        y1 = float(rng.uniform(0, 60)) if kind == 3 else float(rng.uniform(0, 700))
        bbox[i] = (x1, y1, x1 + w, y1 + h)
        score[i] = rng.choice([0.05, 0.1, 0.15, 0.2, 0.5, 0.69, 0.7, 0.71, 0.8, 0.84, 0.9, 0.95, 1.0]) if kind == 4 else rng.uniform(0.02, 1.0)
        spread = float(rng.uniform(2, 90)) if kind != 5 else float(rng.uniform(0.5, 40))
        kps[i] = np.array([x1 + w / 2, y1 + h / 2], np.float32) + rng.uniform(-spread, spread, (5, 2)).astype(np.float32)
        if i % 5 == 0:                             # pose angles available (radians): around the 35-degree thresholds
            pose[i] = np.radians(rng.choice([0.0, 10.0, 34.0, 35.0, 36.0, 60.0, -36.0, -20.0], 2))
    quality = np.zeros((n, 5), np.float64)          # overall, blur, pose, lighting, size
    side = np.zeros(n, np.int32)
    bbox_side = np.zeros((n, 2), np.int32)          # analyze_bbox_for_side_face: flag, score
    for i in range(n):
        face = types.SimpleNamespace(bbox=bbox[i], kps=kps[i], det_score=score[i])
        if pose[i, 0] != 0:
            face.yaw = float(pose[i, 0])
        if pose[i, 1] != 0:
            face.pitch = float(pose[i, 1])
        q = me.assess_face_quality(face)
        quality[i] = (q["overall"], q["blur"], q["pose"], q["lighting"], q["size"])
        side[i] = int(bool(me.is_side_face(face)))
        x1, y1, x2, y2 = bbox[i]
        flag, _, sc = me.analyze_bbox_for_side_face({"width": x2 - x1, "height": y2 - y1, "top": y1, "left": x1}, score[i])
        bbox_side[i] = (int(bool(flag)), int(sc))
    np.savez_compressed(args.out, bbox=bbox, kps=kps, score=score, pose=pose, quality=quality, side=side, bbox_side=bbox_side,
                        config=np.frombuffer(json.dumps({k: cfg[k] for k in ("face_detection", "face_quality", "side_face_detection")}).encode(), np.uint8))
    print(f"{args.out}: {n} faces, {int(side.sum())} side faces, numpy {np.__version__}")


if __name__ == "__main__":
    main()

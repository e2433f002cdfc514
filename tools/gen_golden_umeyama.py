#!/opt/conda/bin/python3.9
"""Golden vectors for estimate_norm (reference utils/helpers.py:18-53).

Run with /opt/conda/bin/python3.9 (the only interpreter here that has scikit-image, 0.18.3):
the reference function runs UNMODIFIED with the real skimage SimilarityTransform; only `cv2`
(not used by estimate_norm) is stubbed.  Output: tests/golden/umeyama.npz
  landmarks f32[n,5,2]  ->  M f64[n,2,3]
"""
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
if not os.path.isdir(ref):
    sys.exit("reference tree not present")
sys.modules["cv2"] = types.ModuleType("cv2")
sys.path.insert(0, ref)
import utils.helpers as H  # noqa: E402  (reference module)

rng = np.random.default_rng(42)
tmpl = H.reference_alignment[0].astype(np.float64)
lms = []
for i in range(96):
    s = rng.uniform(0.3, 6.0)
    th = rng.uniform(-np.pi, np.pi) if i % 3 == 0 else rng.uniform(-0.5, 0.5)
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    t = rng.uniform(0, 1500, size=2)
    pts = (tmpl - 56.0) @ R.T * s + t + rng.normal(0, 1.5 * s, size=(5, 2))
    if i % 11 == 10:                     # mirrored faces: exercises the det(A) < 0 branch
        pts[:, 0] = 2 * t[0] - pts[:, 0]
    lms.append(pts)
for i in range(32):                      # what random detector weights produce: arbitrary points
    lms.append(rng.uniform(-50, 700, size=(5, 2)))
lms.append(np.array([[300, 200], [380, 205], [338, 250], [305, 290], [372, 295]], dtype=np.float64))
lms = np.asarray(lms, dtype=np.float32)
Ms = np.zeros((len(lms), 2, 3), dtype=np.float64)
for i, lm in enumerate(lms):
    M, idx = H.estimate_norm(lm, 112)
    assert idx == 0 and M.dtype == np.float64 and M.shape == (2, 3)
    Ms[i] = M
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "umeyama.npz")
np.savez_compressed(out, landmarks=lms, M=Ms)
print("wrote", out, Ms[-1])

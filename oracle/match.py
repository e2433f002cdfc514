"""Oracle: cosine similarity and the gallery scan.  Follows reference utils/helpers.py:110-123
and main.py:136-142.  Test infrastructure only."""
import numpy as np


def compute_similarity(feat1, feat2):
    feat1 = feat1.ravel()
    feat2 = feat2.ravel()
    return np.dot(feat1, feat2) / (np.linalg.norm(feat1) * np.linalg.norm(feat2))


def gallery_scan(embedding, gallery, similarity_thresh):
    """main.py:136-142 per-target loop: strict '>' against both the running max (starts at 0)
    and the threshold, so the FIRST maximum wins ties.  Returns (index or -1, max_similarity)."""
    max_similarity = 0
    best = -1
    for j in range(len(gallery)):
        s = compute_similarity(gallery[j], embedding)
        if s > max_similarity and s > similarity_thresh:
            max_similarity = s
            best = j
    return best, np.float32(max_similarity)


def match_batch(embeddings, gallery, similarity_thresh):
    """Vectorised equivalent of gallery_scan for many embeddings (fp32 numpy, used at sizes where
    the python loop would take minutes).  argmax returns the first maximum, like the strict '>'."""
    with np.errstate(invalid="ignore", divide="ignore"):
        e = embeddings / np.linalg.norm(embeddings, axis=1, keepdims=True)
        g = gallery / np.linalg.norm(gallery, axis=1, keepdims=True)
        s = e @ g.T
    # an all-zero target or embedding gives NaN similarities; in the reference loop `nan > max_similarity` is False, so such
    # an entry is skipped and the scan goes on (main.py:139-140)
    s = np.where(np.isnan(s), -np.inf, s)
    idx = s.argmax(axis=1)
    best = s[np.arange(len(e)), idx]
    ok = (best > 0) & (best > similarity_thresh)
    return np.where(ok, idx, -1).astype(np.int32), np.where(ok, best, 0).astype(np.float32)

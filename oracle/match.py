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


# ---- the product layer's use of the vector store: reference qdrant_manager.py:137-183, smart_face_recognition.py:2618-2652,2726-2797.
# PARITY UNPINNED for the search itself: qdrant_client is absent; Qdrant's documented behaviour for a cosine collection is restated (vectors are
# normalised on insert, `score_threshold` keeps scores >= the threshold, results best first).  The loops around it follow the reference line by line.

def search_similar(query, ids, embeddings, k=5, threshold=0.0):
    """qdrant_manager.py:137-183: [(id, similarity)] of the k most similar stored embeddings with similarity >= threshold, best first"""
    if len(ids) == 0:
        return []
    e = np.asarray(embeddings, np.float64)
    q = np.asarray(query, np.float64).ravel()
    s = (e / np.linalg.norm(e, axis=1, keepdims=True)) @ (q / np.linalg.norm(q))
    order = np.argsort(-s, kind="stable")[:k]
    return [(ids[j], float(s[j])) for j in order if s[j] >= threshold]


def is_duplicate_embedding(embedding, ids, embeddings, duplicate_threshold=0.95):
    """the vector half of smart_face_recognition.py:2618-2652 (`is_duplicate_image`: the SQL URL checks are storage, out of scope): an embedding
    is a duplicate when its nearest stored neighbour reaches config.json's duplicate_similarity_threshold (0.95)"""
    return len(ids) > 0 and len(search_similar(embedding, ids, embeddings, k=1, threshold=duplicate_threshold)) > 0


def find_and_merge_duplicates(ids, embeddings, similarity_threshold=0.8):
    """smart_face_recognition.py:2726-2797 on the vector store alone (the SQL bookkeeping of merge_duplicate_persons, :2678-2724, is storage): persons
    in ascending id order; each one that still has an embedding searches ALL persons (k = the current person count) above config.json's
    merge_duplicate_threshold (0.8); every hit with a LARGER id that has not been paired yet is merged into it = its embedding is deleted (:2712).
    Returns (merges [(kept id, deleted id, similarity)] in the order they happen, surviving ids)."""
    emb = {i: np.asarray(e, np.float64) for i, e in zip(ids, embeddings)}
    persons = sorted(ids)                        # `SELECT id, name FROM persons ORDER BY id`
    alive = set(persons)                         # ids that still have an embedding in the vector store
    n_persons = len(persons)                     # `persons` is rebound after every merge: k = len(persons) shrinks, the loop walks the original list
    processed, merges = set(), []
    for p1 in persons:
        if p1 not in alive:                      # `embedding1 is None` -> continue
            continue
        live = sorted(alive)
        hits = search_similar(emb[p1], live, [emb[i] for i in live], k=n_persons, threshold=similarity_threshold)
        for p2, sim in hits:
            if p1 >= p2 or (p1, p2) in processed or (p2, p1) in processed:
                continue
            processed.add((p1, p2))
            alive.discard(p2)                    # merge_duplicate_persons -> vector_db.delete_embedding(person_id2)
            merges.append((p1, p2, sim))
            n_persons -= 1
    return merges, sorted(alive)

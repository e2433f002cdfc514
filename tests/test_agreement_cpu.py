"""CPU: the survivor-agreement checker itself (oracle/agreement.py) on synthetic head tensors -- identical heads match
completely, fp16-sized perturbations produce only marginal / cascade flips, a gross change is reported as unexplained."""
import numpy as np

from oracle import agreement as ag


def _heads(rng, hw=(640, 640), n_faces=12):
    """9 session outputs with a few clusters of overlapping high-score candidates (like a detector firing around faces)"""
    outs_s, outs_b, outs_k = [], [], []
    for stride in (8, 16, 32):
        h, w = hw[0] // stride, hw[1] // stride
        n = h * w * 2
        s = rng.uniform(0.0, 0.3, (n, 1)).astype(np.float32)
        b = rng.uniform(1.0, 3.0, (n, 4)).astype(np.float32)
        k = rng.uniform(-1, 1, (n, 10)).astype(np.float32)
        for _ in range(n_faces):
            y, x = rng.integers(2, h - 2), rng.integers(2, w - 2)
            for dy in (0, 1):
                for dx in (0, 1):
                    for a in (0, 1):
                        i = ((y + dy) * w + (x + dx)) * 2 + a
                        s[i, 0] = rng.uniform(0.45, 0.95)
                        b[i] = rng.uniform(2.0, 4.0, 4)
        outs_s.append(s); outs_b.append(b); outs_k.append(k)
    return outs_s + outs_b + outs_k


def test_identical_heads_agree_completely():
    outs = _heads(np.random.default_rng(0))
    r = ag.survivor_agreement(outs, [o.copy() for o in outs])
    assert r["survivors_a"] == r["survivors_b"] == r["matched"] > 10
    assert r["marginal_flips"] == r["cascade_flips"] == r["unexplained"] == 0


def test_small_perturbations_give_only_marginal_flips():
    tot = []
    for seed in range(12):
        rng = np.random.default_rng(seed)
        outs = _heads(rng)
        pert = [o + rng.normal(0, 8e-4, o.shape).astype(np.float32) * (1.0 if i < 3 else 4.0) for i, o in enumerate(outs)]
        tot.append(ag.survivor_agreement(outs, pert))
    s = ag.summarize(tot)
    assert s["unexplained"] == 0, [t["detail"] for t in tot if t["unexplained"]]
    assert s["matched"] > 0.9 * s["survivors_a"]
    assert s["marginal_flips"] > 0                                   # candidates were planted around the threshold: some do flip


def test_gross_change_is_unexplained():
    rng = np.random.default_rng(3)
    outs = _heads(rng)
    det, _ = ag.pp.detect_from_heads(outs, (640, 640), (640, 640), 0.5, 0.4, 0)
    bad = [o.copy() for o in outs]
    # kill the best-scoring ISOLATED survivor's anchor outright (score 0.9 -> 0.0): nothing marginal about that
    sc = np.concatenate([o.ravel() for o in outs[:3]])
    cand = np.argsort(-sc)
    for flat in cand[:40]:
        lvl = 0 if flat < outs[0].size else 1 if flat < outs[0].size + outs[1].size else 2
        off = flat - sum(o.size for o in outs[:lvl])
        if sc[flat] < 0.6:
            continue
        trial = [o.copy() for o in outs]
        trial[lvl][off, 0] = 0.0
        r = ag.survivor_agreement(outs, trial)
        if r["unexplained"] >= 1:
            return
    raise AssertionError("no gross change was reported as unexplained")


def test_top1_agreement_classifies_the_embedded_face():
    """reference scrfd.py:159-177 (max_num = 1: the survivor of largest area is the face main.py:130-134 embeds): identical heads pick the same
    face, fp16-sized perturbations never produce an unexplained pick, growing another face's box by 30 % does"""
    rng = np.random.default_rng(7)
    outs = _heads(rng)
    r = ag.survivor_agreement(outs, [o.copy() for o in outs])
    v, ia, ib = ag.top1_agreement(r)
    assert v == "same" and ia == ib >= 0
    d1, _ = ag.pp.detect_from_heads(outs, (640, 640), (640, 640), 0.5, 0.4, 1)
    assert np.array_equal(d1[0], r["det_a"][ia])                      # the checker's pick IS detect(max_num=1)'s
    verdicts = []
    for seed in range(12):
        rng = np.random.default_rng(100 + seed)
        outs = _heads(rng)
        pert = [o + rng.normal(0, 8e-4, o.shape).astype(np.float32) * (1.0 if i < 3 else 4.0) for i, o in enumerate(outs)]
        verdicts.append(ag.top1_agreement(ag.survivor_agreement(outs, pert))[0])
    assert "unexplained" not in verdicts and verdicts.count("same") >= 9, verdicts
    # gross: on side b another survivor's box grows well past the area of a's pick
    outs = _heads(np.random.default_rng(7))
    r = ag.survivor_agreement(outs, [o.copy() for o in outs])
    _, ia, _ = ag.top1_agreement(r)
    areas = (r["det_a"][:, 2] - r["det_a"][:, 0]) * (r["det_a"][:, 3] - r["det_a"][:, 1])
    other = int(np.argsort(areas)[-3])                                 # a clearly smaller survivor
    assert areas[other] < 0.97 * areas[ia]
    r2 = dict(r)
    det_b = r["det_b"].copy()
    j = int(r["pair_a"][other])
    cx, cy = (det_b[j, 0] + det_b[j, 2]) / 2, (det_b[j, 1] + det_b[j, 3]) / 2
    w, h = (det_b[j, 2] - det_b[j, 0]) * 1.02, (det_b[j, 3] - det_b[j, 1]) * 1.02     # (IoU with its counterpart stays >= 0.9)
    scale = np.sqrt(1.3 * areas[ia] / (w * h))
    r2["det_b"] = det_b
    # keep the pairing (the checker is given one), enlarge the box so that it wins the area arg-max on side b
    det_b[j, :4] = [cx - w * scale / 2, cy - h * scale / 2, cx + w * scale / 2, cy + h * scale / 2]
    v, ia2, ib2 = ag.top1_agreement(r2)
    assert ia2 == ia and ib2 == j and v == "unexplained"


def test_top1_agreement_without_faces():
    e = np.zeros((0, 5), np.float32)
    assert ag.top1_agreement({"det_a": e, "det_b": e, "pair_a": np.zeros(0, int), "pair_b": np.zeros(0, int), "detail": []})[0] == "empty"

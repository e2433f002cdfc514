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

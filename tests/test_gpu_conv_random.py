"""GPU: seeded random layer shapes through every conv kernel family that takes stride-1 layers (register-staged implicit GEMM,
channel-chunked, producer/consumer, resident-weight, two-tile, weights-in-registers conv3x3_wr in its four variants, weights-in-
registers implicit GEMM conv_gw) against the fp32 CPU oracle on the same weights and frames -- odd maps, partial tiles, odd tile
counts, single images.  A family that takes none of the case's layers is not counted (the plan read-back tells)."""
import numpy as np
import pytest

from oracle import align, nets as onets
from scrfd_arcface_facerecognition_amd import archs
from scrfd_arcface_facerecognition_amd.archs import Conv, Net

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from scrfd_arcface_facerecognition_amd._lib import Context
    c = Context(0)
    yield c
    c.close()


def random_case(rng):
    h, w = int(rng.integers(12, 45)), int(rng.integers(12, 45))
    w -= w % 4                                            # the first conv reads aligned dwords of the frame rows
    c1 = int(rng.choice([64, 64, 96, 128, 192, 256]))
    c2 = int(rng.choice([64, 64, 128, 80, 256]))
    batch = int(rng.integers(1, 6))
    acts = [str(rng.choice(["relu", "prelu", "none"])) for _ in range(3)]
    return (h, max(w, 12)), c1, c2, batch, acts, bool(rng.integers(0, 2)), bool(rng.integers(0, 2))


def build(hw, c1, c2, acts, res, pre_bn):
    net = Net("t", hw, 127.5, 1.0 / 128.0)
    net.add(Conv("s", "input", 3, 64, act="relu"))
    net.add(Conv("a", "s", 64, c1, act=acts[0], pre_bn=pre_bn))
    net.add(Conv("b", "a", c1, c1, act=acts[1], res="a" if res else None))
    net.add(Conv("c", "b", c1, c2, act=acts[2]))
    net.outputs = ["c"]
    return net


# (FID_FORCE_GEN, FID_FORCE_NS) per family; conv3x3_wr: NS 1 / 2 = one / two tiles per item, 3 = resident weights, 4 = four-slot ring, 6 / 7 = K split over two wave groups (conv_ks.hip) with one / two items per workgroup
# round 5: 8 = conv_ks on STRIP tiles, 9 / 29 / 39 = conv3x3_wr on STRIP tiles (one tile x 64 couts / a pair x 128 / one tile x 128 on eight waves), 10 = resident weights on STRIP tiles
FAMILIES = [(1, None), (3, None), (5, None), (7, None), (8, None), (9, 1), (9, 2), (9, 3), (9, 4), (9, 6), (9, 7), (9, 8), (9, 9), (9, 29), (9, 39), (9, 10), (11, None)]


def took(plans, gen, ns):
    def hit(p):
        if p["gen"] != gen:
            return False
        if gen != 9 or ns is None:
            return True
        return {1: p["ns"] == 0 and p["bm"] // 256 == 1, 2: p["ns"] == 0 and p["bm"] // 256 == 2, 3: p["ns"] == 1, 4: p["ns"] == 4, 6: p["ns"] == 6 and p["bm"] == 256, 7: p["ns"] == 6 and p["bm"] == 512,
                8: p["ns"] == 7, 9: p["ns"] == 8 and (p["bm"], p["bn"]) == (256, 64), 29: p["ns"] == 8 and p["bm"] == 512, 39: p["ns"] == 8 and (p["bm"], p["bn"]) == (256, 128), 10: p["ns"] == 9}[ns]
    return [p["name"] for p in plans if hit(p)]


@pytest.mark.parametrize("seed", range(24))
def test_families_vs_oracle(ctx, monkeypatch, seed):
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet
    rng = np.random.default_rng(1000 + seed)
    hw, c1, c2, batch, acts, res, pre_bn = random_case(rng)
    net = build(hw, c1, c2, acts, res, pre_bn)
    P = archs.synth_params(net, seed=seed)
    images = rng.integers(0, 256, (batch,) + hw + (3,), dtype=np.uint8)
    ref = onets.run_net(net, P, align.blob_from_images(list(images), net.in_scale, net.in_mean))["c"]
    ref = np.transpose(ref, (0, 2, 3, 1))
    scale = np.abs(ref).max() + 1e-6
    n_checked = 0
    for gen, ns in FAMILIES:
        monkeypatch.setenv("FID_FORCE_GEN", str(gen))
        if ns is None:
            monkeypatch.delenv("FID_FORCE_NS", raising=False)
        else:
            monkeypatch.setenv("FID_FORCE_NS", str(ns))
        cn = CompiledNet(ctx, net, P, max_batch=batch)
        cn.run(images)
        o = cn.read("c", batch).astype(np.float32)
        ran = took(cn.plans(), gen, ns)
        cn.close()
        if not ran:
            continue
        n_checked += 1
        assert np.isfinite(o).all(), (gen, ns, hw, c1, c2, batch)
        assert np.abs(o - ref).max() / scale < 8e-3, (gen, ns, ran, hw, c1, c2, batch, acts, res, pre_bn)
    assert n_checked >= 2, (hw, c1, c2)

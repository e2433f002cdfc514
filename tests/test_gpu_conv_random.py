"""GPU: seeded random layer shapes through the halo-patch kernel families (producer/consumer, two-tile, resident-weight,
channel-chunked) against the register-staged implicit GEMM (generation 1) on the same weights and frames: the families must
agree to fp16 summation-order noise on every shape they accept (odd maps, partial tiles, odd tile counts, single images)."""
import numpy as np
import pytest

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


@pytest.mark.parametrize("seed", range(24))
def test_families_agree(ctx, monkeypatch, seed):
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet
    rng = np.random.default_rng(1000 + seed)
    hw, c1, c2, batch, acts, res, pre_bn = random_case(rng)
    net = build(hw, c1, c2, acts, res, pre_bn)
    P = archs.synth_params(net, seed=seed)
    images = rng.integers(0, 256, (batch,) + hw + (3,), dtype=np.uint8)
    outs = {}
    for gen in (1, 3, 5, 7, 8):
        monkeypatch.setenv("FID_FORCE_GEN", str(gen))
        cn = CompiledNet(ctx, net, P, max_batch=batch)
        cn.run(images)
        outs[gen] = cn.read("c", batch).astype(np.float32)
        cn.close()
    ref = outs[1]
    scale = np.abs(ref).max() + 1e-6
    for gen, o in outs.items():
        assert np.isfinite(o).all(), (gen, hw, c1, c2, batch)
        assert np.abs(o - ref).max() / scale < 4e-3, (gen, hw, c1, c2, batch, acts, res, pre_bn)

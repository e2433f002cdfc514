"""GPU parity on RANDOM shapes: the fused bottleneck (csrc/mbf_block.hip), the 32 / 64-channel fused residual block (csrc/conv_bb.hip) and the
LDS-tiled depthwise kernel (net.hip dwconv3x3_lds) against the fp32 oracle -- odd maps, channel counts that are not multiples of 32, partial
tiles, both strides, more (tile, cout block) items than CUs."""
import os

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


def run(ctx, net, P, images, out):
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet
    cn = CompiledNet(ctx, net, P, max_batch=len(images))
    cn.run(images)
    got = cn.read(out, len(images))
    cn.close()
    ref = np.transpose(onets.run_net(net, P, align.blob_from_images(list(images), net.in_scale, net.in_mean))[out], (0, 2, 3, 1))
    assert got.shape == ref.shape
    return np.abs(got - ref).max() / (np.abs(ref).max() + 1e-6)


@pytest.mark.parametrize("seed", range(int(os.environ.get("FID_FUZZ_SEEDS", "14"))))       # (FID_FUZZ_SEEDS=n: a longer one-off sweep)
def test_random_bottleneck(ctx, seed):
    from scrfd_arcface_facerecognition_amd import lower
    rng = np.random.default_rng(4000 + seed)
    hw = (int(rng.integers(5, 61)), int(rng.integers(5, 61)))
    cin = int(rng.choice([24, 32, 64, 88, 128, 160, 256]))
    g = int(rng.choice([32, 48, 96, 128, 200, 256, 384, 512]))
    stride = int(rng.integers(1, 3))
    res = bool(rng.integers(0, 2)) and stride == 1
    cout = cin if res else int(rng.choice([16, 40, 64, 128, 144, 256]))
    batch = int(rng.choice([1, 2, 3, 11]))
    acts = [str(rng.choice(["prelu", "relu", "none"])) for _ in range(3)]
    net = Net("t", hw, 127.5, 1.0 / 128.0)
    net.add(Conv("s", "input", 3, 64, act="relu"))
    net.add(Conv("x", "s", 64, cin, k=1, pad=0, act="prelu"))
    net.add(Conv("b.pw1", "x", cin, g, k=1, pad=0, act=acts[0]))
    net.add(Conv("b.dw", "b.pw1", g, g, stride=stride, groups=g, act=acts[1]))
    net.add(Conv("b.pw2", "b.dw", g, cout, k=1, pad=0, act=acts[2], res="x" if res else None))
    net.outputs = ["b.pw2"]
    P = archs.synth_params(net, seed=seed)
    fused = sum(int(r[0]) == 8 for r in lower.lower(net, P).ops)
    images = rng.integers(0, 256, (batch,) + hw + (3,), dtype=np.uint8)
    err = run(ctx, net, P, images, "b.pw2")
    # fused exactly when the 9 x 9 input region and both expanded maps fit LDS (csrc/mbf_block.hip mbf_lds_bytes)
    rup, cin_p, gp, to = (lambda v: (v + 1023) // 1024 * 1024), (cin + 31) // 32 * 32, (g + 31) // 32 * 32, (7 if stride == 1 else 4)
    fits = rup(81 * (cin_p * 2 + 16)) + rup(81 * (gp * 2 + 16)) + ((to * to + 15) // 16 * 16) * (gp * 2 + 16) <= 160 * 1024
    assert fused == int(fits), (hw, cin, g, cout, stride, res)
    assert err < 8e-3, (hw, cin, g, cout, stride, res, batch, acts, err)


@pytest.mark.parametrize("seed", range(int(os.environ.get("FID_FUZZ_SEEDS", "8"))))
def test_random_residual_block_and_depthwise(ctx, seed):
    """a fused residual block (32 or 64 stored channels) followed by a stride-1 depthwise layer large enough for the LDS-tiled kernel"""
    rng = np.random.default_rng(5000 + seed)
    hw = (int(rng.integers(12, 120)), int(rng.integers(12, 120)))
    planes = int(rng.choice([16, 24, 32, 40, 56, 64]))
    batch = int(rng.integers(1, 4)) if hw[0] * hw[1] > 4000 else int(rng.integers(8, 14))
    act2 = str(rng.choice(["relu", "none"]))
    net = Net("t", hw, 127.5, 1.0 / 128.0)
    net.add(Conv("s", "input", 3, planes, act="relu"))
    net.add(Conv("b.conv1", "s", planes, planes, act="relu"))
    net.add(Conv("b.conv2", "b.conv1", planes, planes, act=act2, res="s"))
    net.add(Conv("p", "b.conv2", planes, 64, k=1, pad=0, act="prelu"))
    net.add(Conv("d", "p", 64, 64, groups=64, act=str(rng.choice(["prelu", "relu", "none"]))))
    net.outputs = ["d"]
    P = archs.synth_params(net, seed=100 + seed)
    images = rng.integers(0, 256, (batch,) + hw + (3,), dtype=np.uint8)
    err = run(ctx, net, P, images, "d")
    assert err < 8e-3, (hw, planes, batch, act2, err)


@pytest.mark.parametrize("seed", range(int(os.environ.get("FID_FUZZ_SEEDS", "8"))))
def test_random_absorbed_shortcut(ctx, monkeypatch, seed):
    """IResNet's downsampling block with the shortcut conv forced onto conv2's K axis (generation 12): odd maps, 32- and 64-wide K-steps, split-K picks"""
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet
    monkeypatch.setenv("FID_FORCE_GEN", "12")
    rng = np.random.default_rng(6000 + seed)
    hw = (int(rng.integers(6, 70)), int(rng.integers(6, 70)))
    cin = int(rng.choice([64, 88, 128, 200, 256]))
    cout = int(rng.choice([64, 96, 128, 224, 256, 512]))
    batch = int(rng.integers(1, 6))
    net = Net("t", hw, 127.5, 1.0 / 128.0)
    net.add(Conv("s", "input", 3, 64, act="prelu"))
    x = "s"
    if cin != 64:
        net.add(Conv("x", "s", 64, cin, k=1, pad=0, act="prelu"))
        x = "x"
    net.add(Conv("b.down", x, cin, cout, k=1, stride=2, pad=0))
    net.add(Conv("b.conv1", x, cin, cout, act="prelu", pre_bn=True))
    net.add(Conv("b.conv2", "b.conv1", cout, cout, stride=2, res="b.down"))
    net.outputs = ["b.conv2"]
    P = archs.synth_params(net, seed=200 + seed)
    images = rng.integers(0, 256, (batch,) + hw + (3,), dtype=np.uint8)
    cn = CompiledNet(ctx, net, P, max_batch=batch)
    for _ in range(2):
        cn.run(images)
    got = cn.read("b.conv2", batch)
    picks = {p["name"]: (p["gen"], p["ksplit"]) for p in cn.plans()}
    cn.close()
    # generation 2 walks K in steps of 64 channels when conv2's own input allows it, else 32: the shortcut can ride along when its channel count is a multiple of that
    cp = lambda c: (c + 31) // 32 * 32
    bk = 64 if cp(cout) % 64 == 0 else 32
    assert (picks["b.conv2"][0] == 12) == (cp(cin) % bk == 0), (picks, cin, cout)
    ref = np.transpose(onets.run_net(net, P, align.blob_from_images(list(images), net.in_scale, net.in_mean))["b.conv2"], (0, 2, 3, 1))
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() / np.abs(ref).max() < 8e-3, (hw, cin, cout, batch, picks["b.conv2"])


@pytest.mark.parametrize("seed", range(int(os.environ.get("FID_FUZZ_SEEDS", "8"))))
def test_random_absorbed_avg_down_shortcut(ctx, monkeypatch, seed):
    """SCRFD's downsampling BasicBlock (ResNetV1e "avg_down": 2x2 average pool + 1x1 conv + BN shortcut, conv1 3x3 / stride 2, conv2 3x3 / stride 1
    + shortcut, ReLU) with the shortcut forced onto conv2's K axis as FOUR extra taps on the block input (generation 12, round 4), and the same
    block with the shortcut as its own op (fresh autotune without generation 12): both against the fp32 oracle.  Even map sizes only: on an odd
    map conv1 (stride 2, pad 1) yields ceil(H / 2) rows and the average pool floor(H / 2), which is not a valid block."""
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet
    rng = np.random.default_rng(6500 + seed)
    hw = (2 * int(rng.integers(4, 45)), 2 * int(rng.integers(4, 45)))
    cin = int(rng.choice([88, 96, 128, 48, 224]))                 # (64 stored input channels + 96 couts would take the DUAL launch instead)
    cout = int(rng.choice([88, 224, 128, 80, 256]))
    batch = int(rng.integers(1, 6))
    net = Net("t", hw, 127.5, 1.0 / 128.0)
    net.add(Conv("s", "input", 3, 32, act="relu"))
    net.add(Conv("x", "s", 32, cin, k=1, pad=0, act="relu"))
    net.add(Conv("b.down", "x", cin, cout, k=1, stride=1, pad=0, pre_avgpool=True))
    net.add(Conv("b.conv1", "x", cin, cout, stride=2, act="relu"))
    net.add(Conv("b.conv2", "b.conv1", cout, cout, act="relu", res="b.down"))
    net.outputs = ["b.conv2"]
    P = archs.synth_params(net, seed=300 + seed)
    images = rng.integers(0, 256, (batch,) + hw + (3,), dtype=np.uint8)
    ref = np.transpose(onets.run_net(net, P, align.blob_from_images(list(images), net.in_scale, net.in_mean))["b.conv2"], (0, 2, 3, 1))
    cp = lambda c: (c + 31) // 32 * 32
    for force in ("12", None):
        if force:
            monkeypatch.setenv("FID_FORCE_GEN", force)
        else:
            monkeypatch.delenv("FID_FORCE_GEN")
            monkeypatch.setenv("FID_NO_SC_FUSE", "1")             # (lower.py: no second weight image, the shortcut stays an op of its own)
        cn = CompiledNet(ctx, net, P, max_batch=batch)
        for _ in range(2):
            cn.run(images)
        got = cn.read("b.conv2", batch)
        picks = {p["name"]: (p["gen"], p["ksplit"]) for p in cn.plans()}
        cn.close()
        bk = 64 if cp(cout) % 64 == 0 else 32
        if force:
            assert (picks["b.conv2"][0] == 12) == (cp(cin) % bk == 0), (picks, cin, cout)
        else:
            assert picks["b.conv2"][0] != 12
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() / np.abs(ref).max() < 8e-3, (hw, cin, cout, batch, force, picks["b.conv2"])

"""GPU parity: every conv kernel family (direct, gen-1 igemm, gen-2 LDS-DMA ring, channel-chunked direct, ping-pong chunked, producer/consumer chunked, producer/consumer implicit GEMM, producer/consumer resident-weight 64-channel conv),
forced through the autotuner hook FID_FORCE_GEN, against the fp32 oracle on the same layer stacks."""
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


def forced_ran(cn, code):
    """ops of the net whose pick is the kernel family the test forced (code = the test's generation code: FID_FORCE_GEN, or
    FID_FORCE_GEN * 10 + FID_FORCE_NS for the ring / tile variants).  FID_FORCE_GEN only restricts the candidates where the family
    applies; a test that asserted nothing about the pick would pass on another family's result."""
    def hit(p):
        if code in (25, 51):
            return p["gen"] == code // 10 and p["ns"] == (5 if code == 25 else 1)
        if code == 59:
            return p["gen"] == 5
        if code in (91, 92):
            return p["gen"] == 9 and p["ns"] not in (1, 4, 6) and p["bm"] // 256 == code % 10
        if code == 93:
            return p["gen"] == 9 and p["ns"] == 1
        if code == 94:
            return p["gen"] == 9 and p["ns"] == 4
        if code in (96, 97):                                    # conv_ks: one / two items per workgroup (plan tile 256 / 512)
            return p["gen"] == 9 and p["ns"] == 6 and p["bm"] == (256 if code == 96 else 512)
        if code == 98:                                          # conv_ks on x-packed STRIP tiles
            return p["gen"] == 9 and p["ns"] == 7
        if code in (909, 929, 939):                             # conv_wr on STRIP tiles: one tile x 64 couts / a pair x 128 couts / one tile x 128 couts
            return p["gen"] == 9 and p["ns"] == 8 and (p["bm"], p["bn"]) == {909: (256, 64), 929: (512, 128), 939: (256, 128)}[code]
        if code == 910:                                         # conv_wr on STRIP tiles, the layer's weights resident
            return p["gen"] == 9 and p["ns"] == 9
        return p["gen"] == code
    return [p["name"] for p in cn.plans() if hit(p)]


def stack(hw, chans, res=True):
    net = Net("t", hw, 127.5, 1.0 / 128.0)
    net.add(Conv("s", "input", 3, 64, act="relu"))
    src, cin = "s", 64
    for i, c in enumerate(chans):
        net.add(Conv(f"a{i}", src, cin, c, act="prelu", pre_bn=(i == 1)))
        net.add(Conv(f"b{i}", f"a{i}", c, c, act="relu", res=f"a{i}" if res else None))
        src, cin = f"b{i}", c
    net.outputs = [src]
    return net


# 25 = generation 2 with ns = 5 (fragment prefetch across K-steps), 51 = generation 5 with the weights two steps ahead,
# 59 = generation 5 with register-staged producers (FID_PC_RS); 8 = two tiles per weight chunk (the (14, 14) x 5 case: an odd tile count);
# 91 / 92 = generation 9 (weights in registers; 14-row tiles on the 28 / 14-pixel maps, 10-row tiles on the 20-pixel maps, 16-row tiles elsewhere) with one tile x 64 couts /
# a pair of tiles x 128 couts per item / (93) one tile x all couts with the layer's weights resident in registers (64 / 96 / 128-cout layers of 64 / 96 channels) / (94) one tile x 64 couts with a four-slot patch ring (pieces three steps ahead); 11 = implicit GEMM with the weights in registers (conv_gw: every conv with >= 96 couts, any kernel size / stride)
# 96 = generation 9 with the K axis split over two wave groups (conv_ks.hip: one tile x 64 couts per item, 8 waves; layers with an even number of 32-channel chunks);
# 97 = the same with two items per workgroup (offered when there are no more items than CUs: half the workgroups, the persistent item loop)
# 98 = conv_ks on x-packed STRIP tiles (round 5): a tile is 16 consecutive columns of the strip of all images' rows (the (14, 14) x 5 case: 70 strip
#      columns = four full tiles and one of six lanes; (37, 21): every one of 21 boundary positions inside a tile)
# 909 / 929 / 910 = conv_wr on STRIP tiles (one tile x 64 couts / a pair of tiles x 128 couts / the resident-weight variant)
@pytest.mark.parametrize("gen", [0, 1, 2, 3, 4, 5, 6, 7, 8, 91, 92, 93, 94, 96, 97, 98, 909, 929, 939, 910, 11, 25, 51, 59])
@pytest.mark.parametrize("hw,chans,batch", [((32, 48), (64, 96), 3), ((28, 28), (128, 256), 5), ((40, 24), (88, 224), 2), ((37, 21), (64, 64), 3),
                                            ((14, 14), (128, 128), 5), ((20, 20), (64, 96), 4), ((20, 20), (224, 224), 3)])
def test_conv_family(ctx, monkeypatch, gen, hw, chans, batch):
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet
    if gen == 59:
        monkeypatch.setenv("FID_FORCE_GEN", "5")
        monkeypatch.setenv("FID_PC_RS", "1")
    elif gen >= 900:
        monkeypatch.setenv("FID_FORCE_GEN", str(gen // 100))
        monkeypatch.setenv("FID_FORCE_NS", str(gen % 100))
    elif gen in (25, 51, 91, 92, 93, 94, 96, 97, 98):
        monkeypatch.setenv("FID_FORCE_GEN", str(gen // 10))
        monkeypatch.setenv("FID_FORCE_NS", str(gen % 10))
    else:
        monkeypatch.setenv("FID_FORCE_GEN", str(gen))
    net = stack(hw, chans)
    P = archs.synth_params(net, seed=9)
    images = np.random.default_rng(3).integers(0, 256, (batch,) + hw + (3,), dtype=np.uint8)
    cn = CompiledNet(ctx, net, P, max_batch=batch)
    cn.run(images)
    got = cn.read(net.outputs[0], batch)
    ran = forced_ran(cn, gen)
    cn.close()
    if not ran:
        pytest.skip(f"generation code {gen} takes no layer of this stack")
    ref = onets.run_net(net, P, align.blob_from_images(list(images), net.in_scale, net.in_mean))[net.outputs[0]]
    ref = np.transpose(ref, (0, 2, 3, 1))
    assert np.abs(got - ref).max() / np.abs(ref).max() < 8e-3, ran


# 7x7 maps (IResNet's last stage): conv_ks packs four images into one 16x16 tile with zero gutters between them (MOSAIC) -- image counts that
# are / are not multiples of four, a single image, plain / border-class bias, PReLU, residual; against the oracle and against conv_gw (11)
@pytest.mark.parametrize("gen", [96, 97, 11])
@pytest.mark.parametrize("chans,batch", [((128, 128), 9), ((64, 192), 3), ((256, 64), 1), ((128, 512), 64), ((64, 64), 4)])
def test_conv_mosaic_7x7(ctx, monkeypatch, gen, chans, batch):
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet
    monkeypatch.setenv("FID_FORCE_GEN", str(gen // 10) if gen > 11 else str(gen))
    if gen > 11:
        monkeypatch.setenv("FID_FORCE_NS", str(gen % 10))
    net = stack((7, 7), chans)
    P = archs.synth_params(net, seed=31)
    images = np.random.default_rng(33).integers(0, 256, (batch, 7, 7, 3), dtype=np.uint8)
    cn = CompiledNet(ctx, net, P, max_batch=batch)
    cn.run(images)
    got = cn.read(net.outputs[0], batch)
    ran = forced_ran(cn, gen)
    cn.close()
    if gen == 97 and not ran:
        pytest.skip("a single item: nothing to pair")
    if gen in (96, 97):
        assert len(ran) >= (3 if gen == 96 else 1), ran                 # every 3x3 conv of the stack but the 3-channel one (and a1 after 64 -> 192: still even chunk counts)
    elif not ran:
        pytest.skip("conv_gw takes no layer of this stack")
    ref = onets.run_net(net, P, align.blob_from_images(list(images), net.in_scale, net.in_mean))[net.outputs[0]]
    ref = np.transpose(ref, (0, 2, 3, 1))
    assert np.abs(got - ref).max() / np.abs(ref).max() < 8e-3, ran


# STRIP tiles (round 5): x-packed pixel fragments -- image counts that fill / do not fill the last tile, a single image, maps of one / two / three
# tile rows, every lane position of the image boundary (21-wide rows), plain / border-class bias, PReLU, residual; against the oracle
@pytest.mark.parametrize("gen", [98, 909, 929, 939, 910])
@pytest.mark.parametrize("hw,chans,batch", [((14, 14), (128, 128), 8), ((14, 14), (64, 256), 9), ((14, 14), (128, 64), 1), ((14, 14), (64, 128), 17),
                                            ((28, 28), (128, 128), 3), ((20, 20), (64, 192), 5), ((12, 12), (64, 64), 6), ((15, 15), (128, 64), 4),
                                            ((37, 21), (64, 128), 3), ((40, 40), (64, 64), 2), ((56, 56), (64, 128), 3), ((40, 40), (96, 96), 3),
                                            ((20, 20), (96, 64), 5), ((20, 20), (224, 224), 3)])
def test_conv_strip(ctx, monkeypatch, gen, hw, chans, batch):
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet
    monkeypatch.setenv("FID_FORCE_GEN", str(gen // (100 if gen >= 900 else 10)))
    monkeypatch.setenv("FID_FORCE_NS", str(gen % (100 if gen >= 900 else 10)))
    net = stack(hw, chans)
    P = archs.synth_params(net, seed=41)
    images = np.random.default_rng(43).integers(0, 256, (batch,) + hw + (3,), dtype=np.uint8)
    cn = CompiledNet(ctx, net, P, max_batch=batch)
    cn.run(images)
    got = cn.read(net.outputs[0], batch)
    ran = forced_ran(cn, gen)
    cn.close()
    if gen == 910:                                         # resident weights: 64 / 96-channel layers with at most 128 couts
        if not ran:
            pytest.skip("no layer of this stack keeps its weights resident")
    elif gen == 98 and any(c % 64 for c in chans):
        if not ran:
            pytest.skip("conv_ks needs an even number of 32-channel chunks")
    elif gen == 939:
        if max(chans) <= 64:
            pytest.skip("one tile x 128 couts: layers with more than 64 couts")
        assert len(ran) >= 1, ran
    else:
        assert len(ran) >= 2, ran                          # at least the convs on the second channel count
    ref = onets.run_net(net, P, align.blob_from_images(list(images), net.in_scale, net.in_mean))[net.outputs[0]]
    ref = np.transpose(ref, (0, 2, 3, 1))
    assert np.abs(got - ref).max() / np.abs(ref).max() < 8e-3, ran


# the detector head maps (32 fp32 channels, sigmoid on the class scores) through every family that takes them
@pytest.mark.parametrize("gen", [0, 1, 2, 5])
@pytest.mark.parametrize("hw,cin,batch", [((96, 160), 80, 2), ((104, 104), 64, 3)])
def test_dethead_family(ctx, monkeypatch, gen, hw, cin, batch):
    from scrfd_arcface_facerecognition_amd.archs import DetHead
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet
    monkeypatch.setenv("FID_FORCE_GEN", str(gen))
    net = Net("t", hw, 127.5, 1.0 / 128.0)
    net.add(Conv("s", "input", 3, 64, stride=2, act="relu"))
    net.add(Conv("t0", "s", 64, cin, stride=2, act="relu"))
    net.add(DetHead("h", "t0", cin, 8))
    net.outputs = ["h"]
    P = archs.synth_params(net, seed=4)
    images = np.random.default_rng(5).integers(0, 256, (batch,) + hw + (3,), dtype=np.uint8)
    cn = CompiledNet(ctx, net, P, max_batch=batch)
    cn.run(images)
    fused = cn.read("h", batch)                          # [B,H,W,30]: cls(2) bbox(8) kps(20)
    ran = forced_ran(cn, gen)
    cn.close()
    if "h" not in ran:
        pytest.skip(f"generation {gen} does not take the head conv")
    sc, bb, kp = onets.run_net(net, P, align.blob_from_images(list(images), net.in_scale, net.in_mean))["h"]
    assert np.abs(fused[..., 0:2].reshape(batch, -1, 1) - sc).max() < 2e-3
    assert np.abs(fused[..., 2:10].reshape(batch, -1, 4) - bb).max() < 2e-2
    assert np.abs(fused[..., 10:30].reshape(batch, -1, 10) - kp).max() < 2e-2


# stride-2 3x3 convs (generation 10: parity-plane patches, resident weights) against the implicit-GEMM families and the oracle:
# 64 / 96 / 128-channel inputs, 64 / 96 / 128 couts, with and without a residual (the IResNet block's downsample branch), odd maps
@pytest.mark.parametrize("gen", [1, 2, 10, 11])
@pytest.mark.parametrize("hw,cin,cout,res,batch", [((64, 96), 64, 64, True, 3), ((37, 45), 64, 96, False, 2), ((80, 80), 88, 88, False, 2),
                                                   ((56, 56), 64, 128, True, 5), ((30, 18), 96, 64, True, 3),
                                                   ((56, 56), 128, 128, True, 5), ((45, 38), 128, 128, False, 2), ((16, 16), 128, 128, True, 1)])   # (round 4: 128 input channels)
def test_stride2_family(ctx, monkeypatch, gen, hw, cin, cout, res, batch):
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet
    monkeypatch.setenv("FID_FORCE_GEN", str(gen))
    net = Net("t", hw, 127.5, 1.0 / 128.0)
    if cin == 64:
        net.add(Conv("s", "input", 3, cin, act="relu"))
    else:                                   # the first conv of a net has 16 / 32 / 64 couts: widen with a second one
        net.add(Conv("s0", "input", 3, 64, act="relu"))
        net.add(Conv("s", "s0", 64, cin, act="relu"))
    if res:
        net.add(Conv("d", "s", cin, cout, k=1, stride=2, pad=0))
        net.add(Conv("c", "s", cin, cout, stride=2, act="prelu", res="d"))
    else:
        net.add(Conv("c", "s", cin, cout, stride=2, act="relu"))
    net.outputs = ["c"]
    P = archs.synth_params(net, seed=11)
    images = np.random.default_rng(4).integers(0, 256, (batch,) + hw + (3,), dtype=np.uint8)
    cn = CompiledNet(ctx, net, P, max_batch=batch)
    cn.run(images)
    got = cn.read("c", batch)
    ran = forced_ran(cn, gen)
    cn.close()
    if "c" not in ran:
        pytest.skip(f"generation {gen} does not take this stride-2 conv")
    ref = onets.run_net(net, P, align.blob_from_images(list(images), net.in_scale, net.in_mean))["c"]
    ref = np.transpose(ref, (0, 2, 3, 1))
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() / np.abs(ref).max() < 6e-3


# the block shortcut (2x2 average pool + 1x1 conv) fused into the stride-2 conv that reads the same tensor (lower.py; conv_s2.hip DUAL):
# fused and unfused lowering against the oracle, even and odd maps (partial tiles), with the conv2 that consumes both outputs
@pytest.mark.parametrize("fuse", [True, False])
@pytest.mark.parametrize("hw,planes,batch", [((64, 96), 88, 3), ((74, 50), 96, 2), ((34, 46), 88, 1)])
def test_fused_shortcut_stride2(ctx, monkeypatch, fuse, hw, planes, batch):
    from scrfd_arcface_facerecognition_amd import lower
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet
    if not fuse:
        monkeypatch.setenv("FID_NO_DOWN_FUSE", "1")
    net = Net("t", hw, 127.5, 1.0 / 128.0)
    net.add(Conv("s", "input", 3, 64, act="relu"))
    net.add(Conv("b.down", "s", 64, planes, k=1, stride=1, pad=0, pre_avgpool=True))
    net.add(Conv("b.conv1", "s", 64, planes, stride=2, act="relu"))
    net.add(Conv("b.conv2", "b.conv1", planes, planes, act="relu", res="b.down"))
    net.outputs = ["b.conv2"]
    P = archs.synth_params(net, seed=21)
    low = lower.lower(net, P)
    assert (len(low.ops) == 3) == fuse and any(int(r[20]) > 0 for r in low.ops if int(r[0]) == 2) == fuse
    images = np.random.default_rng(8).integers(0, 256, (batch,) + hw + (3,), dtype=np.uint8)
    cn = CompiledNet(ctx, net, P, max_batch=batch)
    cn.run(images)
    got = {nm: cn.read(nm, batch) for nm in ("b.down", "b.conv1", "b.conv2")}
    cn.close()
    ref = onets.run_net(net, P, align.blob_from_images(list(images), net.in_scale, net.in_mean), keep=("b.down", "b.conv1", "b.conv2"))
    for nm in got:
        r = np.transpose(ref[nm], (0, 2, 3, 1))
        assert np.abs(got[nm] - r).max() / np.abs(r).max() < 8e-3, nm


# a residual BasicBlock on 64 stored channels as ONE launch (lower.py pattern; conv_bb.hip: 14x14 output tiles, the intermediate map in LDS,
# the residual from the input patch): fused and unfused lowering against the oracle -- maps that are / are not multiples of 14, maps smaller
# than a tile, one image, two chained blocks (the second walks its items in the other direction), both activations after the add
@pytest.mark.parametrize("fuse", [True, "v2", False])      # "v2": conv_bb2 (two four-wave workgroups per CU, the tile finished in place; FID_BB_V=2)
# ("ir": IResNet's form -- BN - conv - BN - PReLU - conv - BN, + input -- conv1 then carries 9 border-class bias rows and PReLU slopes)
@pytest.mark.parametrize("hw,planes,batch,act2", [((56, 84), 56, 3, "relu"), ((37, 45), 64, 2, "relu"), ((12, 20), 56, 5, "none"), ((160, 160), 56, 1, "relu"),
                                                  ((29, 16), 40, 4, "relu"), ((56, 56), 64, 3, "ir"), ((23, 31), 64, 2, "ir"), ((14, 14), 64, 5, "ir"),
                                                  # 32 stored channels (conv_bb32: SCRFD-2.5G layer1): four row groups, rows 14 / 15 of the last one dropped
                                                  ((56, 84), 24, 3, "relu"), ((37, 45), 32, 2, "none"), ((160, 160), 24, 1, "relu"), ((12, 20), 16, 5, "relu")])
def test_fused_basic_block(ctx, monkeypatch, fuse, hw, planes, batch, act2):
    from scrfd_arcface_facerecognition_amd import lower
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet
    if not fuse:
        monkeypatch.setenv("FID_NO_BB_FUSE", "1")
    if fuse == "v2":
        if planes <= 32:
            pytest.skip("conv_bb2 is the 64-channel kernel")
        monkeypatch.setenv("FID_BB_V", "2")
    else:
        monkeypatch.delenv("FID_BB_V", raising=False)
    fuse = bool(fuse)
    net = Net("t", hw, 127.5, 1.0 / 128.0)
    ir = act2 == "ir"
    a1 = dict(act="prelu", pre_bn=True) if ir else dict(act="relu")
    net.add(Conv("s", "input", 3, planes, act="prelu" if ir else "relu"))
    net.add(Conv("b0.conv1", "s", planes, planes, **a1))
    net.add(Conv("b0.conv2", "b0.conv1", planes, planes, act="none" if ir else act2, res="s"))
    net.add(Conv("b1.conv1", "b0.conv2", planes, planes, **a1))
    net.add(Conv("b1.conv2", "b1.conv1", planes, planes, act="none" if ir else "relu", res="b0.conv2"))
    net.outputs = ["b1.conv2"]
    P = archs.synth_params(net, seed=23)
    low = lower.lower(net, P)
    assert (len(low.ops) == 3) == fuse and (sum(int(r[0]) == 6 for r in low.ops) == 2) == fuse
    images = np.random.default_rng(9).integers(0, 256, (batch,) + hw + (3,), dtype=np.uint8)
    cn = CompiledNet(ctx, net, P, max_batch=batch)
    cn.run(images)
    cn.run(images)                                      # twice: the second run starts from the other walking direction state
    got = {nm: cn.read(nm, batch) for nm in ("b0.conv2", "b1.conv2")}
    cn.close()
    ref = onets.run_net(net, P, align.blob_from_images(list(images), net.in_scale, net.in_mean), keep=("b0.conv2", "b1.conv2"))
    for nm in got:
        r = np.transpose(ref[nm], (0, 2, 3, 1))
        assert got[nm].shape == r.shape
        assert np.abs(got[nm] - r).max() / np.abs(r).max() < 8e-3, nm


# depthwise 3x3 + the pointwise 1x1 that consumes it as ONE launch (lower.py pattern; csrc/dwpw.hip: the depthwise result stays in LDS): fused
# and unfused lowering against the oracle -- MobileFaceNet's bottleneck shapes (stride 1 with a residual, stride 2 without), odd maps whose
# pixel count is no multiple of the 64-pixel item, a single image, every activation pair
@pytest.mark.parametrize("fuse", [True, False])
@pytest.mark.parametrize("hw,groups,cout,stride,res,acts,batch", [((28, 28), 128, 128, 1, True, ("prelu", "none"), 3), ((28, 28), 256, 256, 2, False, ("prelu", "none"), 2),
                                                                   ((14, 14), 512, 256, 2, False, ("prelu", "prelu"), 5), ((37, 21), 64, 96, 1, False, ("relu", "relu"), 1),
                                                                   ((16, 24), 32, 48, 2, False, ("none", "relu"), 4)])
def test_fused_depthwise_pointwise(ctx, monkeypatch, fuse, hw, groups, cout, stride, res, acts, batch):
    from scrfd_arcface_facerecognition_amd import lower
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet
    monkeypatch.setenv("FID_NO_MBF_FUSE", "1")                  # (the whole-bottleneck fusion would take the 1x1 in front of the pair as well)
    if fuse:
        monkeypatch.setenv("FID_DWPW_FUSE", "1")                 # (opt-in: measured slower than the two launches on MobileFaceNet)
    else:
        monkeypatch.delenv("FID_DWPW_FUSE", raising=False)
    net = Net("t", hw, 127.5, 1.0 / 128.0)
    net.add(Conv("s", "input", 3, 64, act="relu"))
    net.add(Conv("p1", "s", 64, groups, k=1, pad=0, act="prelu"))
    if res:
        assert cout == groups
        net.add(Conv("p0", "p1", groups, cout, k=1, pad=0))
        src = "p0"
        net.add(Conv("q1", src, cout, groups, k=1, pad=0, act="prelu"))
        net.add(Conv("dw", "q1", groups, groups, stride=stride, groups=groups, act=acts[0]))
        net.add(Conv("pw", "dw", groups, cout, k=1, pad=0, act=acts[1], res=src))
    else:
        net.add(Conv("dw", "p1", groups, groups, stride=stride, groups=groups, act=acts[0]))
        net.add(Conv("pw", "dw", groups, cout, k=1, pad=0, act=acts[1]))
    net.outputs = ["pw"]
    P = archs.synth_params(net, seed=31)
    low = lower.lower(net, P)
    assert (sum(int(r[0]) == 7 for r in low.ops) == 1) == fuse and (sum(int(r[0]) == 4 for r in low.ops) == 0) == fuse
    images = np.random.default_rng(10).integers(0, 256, (batch,) + hw + (3,), dtype=np.uint8)
    cn = CompiledNet(ctx, net, P, max_batch=batch)
    cn.run(images)
    got = cn.read("pw", batch)
    cn.close()
    ref = np.transpose(onets.run_net(net, P, align.blob_from_images(list(images), net.in_scale, net.in_mean))["pw"], (0, 2, 3, 1))
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() / np.abs(ref).max() < 8e-3


# MobileFaceNet's bottleneck (1x1 -> depthwise 3x3 -> 1x1 [+ block input]) as ONE launch with both expanded maps in LDS (csrc/mbf_block.hip):
# fused and unfused lowering against the oracle -- the net's five block shapes (tiled 28x28 / 56x56 maps with a halo, whole 14x14 / 7x7 maps with
# 256 input channels, stride 2 with 512 expanded channels), odd maps with partial tiles, a single image, more than 128 couts (two items per tile),
# ReLU / no activation variants
@pytest.mark.parametrize("fuse", [True, False])
@pytest.mark.parametrize("hw,cin,g,cout,stride,res,acts,batch", [
    ((28, 28), 128, 128, 128, 1, True, ("prelu", "prelu", "none"), 3), ((56, 56), 128, 128, 128, 2, False, ("prelu", "prelu", "none"), 2),
    ((28, 28), 128, 256, 256, 2, False, ("prelu", "prelu", "none"), 2), ((14, 14), 256, 256, 256, 1, True, ("prelu", "prelu", "none"), 5),
    ((14, 14), 256, 512, 256, 2, False, ("prelu", "prelu", "none"), 3), ((7, 7), 256, 256, 256, 1, True, ("prelu", "prelu", "none"), 4),
    ((30, 22), 64, 96, 64, 1, True, ("relu", "relu", "relu"), 1), ((37, 21), 64, 128, 96, 2, False, ("none", "relu", "prelu"), 2),
    ((12, 9), 224, 160, 208, 1, False, ("prelu", "none", "none"), 3),
    # more (tile, cout block) items than CUs: a workgroup walks both cout blocks of its tile itself
    ((28, 28), 128, 256, 256, 2, False, ("prelu", "prelu", "none"), 9), ((21, 30), 96, 128, 160, 1, False, ("prelu", "prelu", "none"), 12)])
def test_fused_bottleneck(ctx, monkeypatch, fuse, hw, cin, g, cout, stride, res, acts, batch):
    from scrfd_arcface_facerecognition_amd import lower
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet
    if not fuse:
        monkeypatch.setenv("FID_NO_MBF_FUSE", "1")
    net = Net("t", hw, 127.5, 1.0 / 128.0)
    net.add(Conv("s", "input", 3, 64, act="relu"))
    net.add(Conv("x", "s", 64, cin, k=1, pad=0, act="prelu"))
    net.add(Conv("b.pw1", "x", cin, g, k=1, pad=0, act=acts[0]))
    net.add(Conv("b.dw", "b.pw1", g, g, stride=stride, groups=g, act=acts[1]))
    net.add(Conv("b.pw2", "b.dw", g, cout, k=1, pad=0, act=acts[2], res="x" if res else None))
    net.outputs = ["b.pw2"]
    P = archs.synth_params(net, seed=51)
    low = lower.lower(net, P)
    assert (sum(int(r[0]) == 8 for r in low.ops) == 1) == fuse and (sum(int(r[0]) == 4 for r in low.ops) == 0) == fuse
    images = np.random.default_rng(12).integers(0, 256, (batch,) + hw + (3,), dtype=np.uint8)
    cn = CompiledNet(ctx, net, P, max_batch=batch)
    cn.run(images)
    got = cn.read("b.pw2", batch)
    cn.close()
    ref = np.transpose(onets.run_net(net, P, align.blob_from_images(list(images), net.in_scale, net.in_mean))["b.pw2"], (0, 2, 3, 1))
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() / np.abs(ref).max() < 8e-3


# IResNet's downsampling block: the shortcut (1x1 / stride-2 conv + BN on the block input) as extra K-steps of the stride-2 conv2 that adds it
# (lower.py keeps both forms in the table, the autotuner decides per batch size; conv.hip generation 2 with a second input tensor = a generation-12
# pick of conv2, the shortcut op is then skipped).  "fused": generation 12 forced; "tuned": whatever the tuner picks; "plain": the lowering without
# the second weight image -- all against the oracle: IResNet-50's four shapes scaled down, odd maps, channel counts whose K-step must be 32 wide
# "fused_s2" (round 4): the same form inside the parity-plane stride-2 kernel (conv_s2.hip NX = 2: one more step per item on the 8 x 16 sampled
# pixels of the block input; a generation-12 pick with ns = 10) -- 64 -> 64 and 128 -> 128 channels with a 64-channel block input, tile-multiple
# and ragged maps (Ho / Wo not multiples of 8 / 16)
@pytest.mark.parametrize("mode", ["fused", "fused_s2", "tuned", "plain"])
@pytest.mark.parametrize("hw,cin,cout,batch", [((56, 56), 64, 64, 2), ((28, 28), 64, 128, 3), ((37, 45), 64, 96, 2), ((14, 14), 256, 512, 5), ((30, 22), 88, 160, 1),
                                               ((44, 70), 64, 64, 3), ((20, 36), 64, 128, 2), ((112, 112), 64, 64, 1), ((64, 32), 64, 128, 5)])
def test_fused_shortcut_conv(ctx, monkeypatch, mode, hw, cin, cout, batch):
    from scrfd_arcface_facerecognition_amd import lower
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet
    if mode == "plain":
        monkeypatch.setenv("FID_NO_SC_FUSE", "1")
    if mode in ("fused", "fused_s2"):
        monkeypatch.setenv("FID_FORCE_GEN", "12")            # (only conv2 has generation-12 candidates: every other op tunes as usual)
    if mode == "fused_s2":
        if not (cin == 64 and cout in (64, 128) and (hw[0] + 1) // 2 >= 8 and (hw[1] + 1) // 2 >= 8):
            pytest.skip("conv3x3_s2's shortcut-absorbing form takes 64 -> 64 and 128 -> 128 channels with a 64-channel block input")
        monkeypatch.setenv("FID_FORCE_NS", "10")
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
    P = archs.synth_params(net, seed=61)
    low = lower.lower(net, P)
    assert (sum(int(r[0]) == 2 and int(r[23]) > 0 for r in low.ops) == 1) == (mode != "plain") and "b.down" in low.op_names
    images = np.random.default_rng(14).integers(0, 256, (batch,) + hw + (3,), dtype=np.uint8)
    cn = CompiledNet(ctx, net, P, max_batch=batch)
    for _ in range(3):                                      # the first run tunes (the shortcut op runs); later runs skip it when conv2's pick is generation 12
        cn.run(images)
    got = cn.read("b.conv2", batch)
    picks = {p["name"]: p["gen"] for p in cn.plans()}
    ns = {p["name"]: p["ns"] for p in cn.plans()}
    cn.close()
    if mode in ("fused", "fused_s2"):
        assert picks["b.conv2"] == 12
    if mode == "fused_s2":
        assert ns["b.conv2"] == 10
    if mode == "plain":
        assert picks["b.conv2"] != 12
    ref = np.transpose(onets.run_net(net, P, align.blob_from_images(list(images), net.in_scale, net.in_mean))["b.conv2"], (0, 2, 3, 1))
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() / np.abs(ref).max() < 8e-3, picks


# IResNet's stem + layer1.0.conv1 as one launch (csrc/stem_block.hip, round 4): the first conv's map never leaves the CU, the block's stride-2
# shortcut reads a compact even-pixel copy of it.  Fused and unfused (FID_NO_STEMBLOCK_FUSE=1) lowering against the oracle: tile-multiple and
# ragged maps (16 x 16 tiles), odd heights, with / without the BatchNorm in front of the second conv (9 / 1 bias rows), ReLU / PReLU, with and
# without the shortcut consumer, and the whole downsampling block behind it (conv2 absorbing the shortcut from the even-pixel copy)
@pytest.mark.parametrize("fuse", [True, False])
@pytest.mark.parametrize("hw,pre_bn,act,block,batch", [((112, 112), True, "prelu", True, 2), ((48, 36), True, "prelu", True, 3), ((50, 44), False, "relu", True, 2),
                                                       ((17, 20), True, "prelu", False, 1), ((33, 64), True, "relu", True, 5), ((16, 16), False, "prelu", True, 1)])
def test_fused_stem_block(ctx, monkeypatch, fuse, hw, pre_bn, act, block, batch):
    from scrfd_arcface_facerecognition_amd import lower
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet
    if not fuse:
        monkeypatch.setenv("FID_NO_STEMBLOCK_FUSE", "1")
    net = Net("t", hw, 127.5, 1.0 / 127.5)
    net.add(Conv("stem", "input", 3, 64, act=act))
    if block:
        net.add(Conv("b.down", "stem", 64, 64, k=1, stride=2, pad=0))
    net.add(Conv("b.conv1", "stem", 64, 64, act=act, pre_bn=pre_bn))
    outs = ["b.conv1"]
    if block:
        net.add(Conv("b.conv2", "b.conv1", 64, 64, stride=2, res="b.down"))
        outs.append("b.conv2")
    net.outputs = outs
    P = archs.synth_params(net, seed=71)
    low = lower.lower(net, P)
    assert (int(low.ops[0][0]) == lower.OP_STEMBLOCK) == fuse
    if fuse:
        assert ("stem.even" in low.tensor_id) == block and "stem" not in low.tensor_id
    images = np.random.default_rng(15).integers(0, 256, (batch,) + hw + (3,), dtype=np.uint8)
    cn = CompiledNet(ctx, net, P, max_batch=batch)
    for _ in range(2):
        cn.run(images)
    got = {o: cn.read(o, batch) for o in outs}
    if block and fuse:                                        # the compact copy = the oracle's stem map at the even pixels
        got["stem.even"] = cn.read("stem.even", batch)
    cn.close()
    ref = onets.run_net(net, P, align.blob_from_images(list(images), net.in_scale, net.in_mean), keep=["stem"])
    for o in outs:
        r = np.transpose(ref[o], (0, 2, 3, 1))
        assert got[o].shape == r.shape
        assert np.abs(got[o] - r).max() / np.abs(r).max() < 6e-3, (o, hw)
    if "stem.even" in got:
        r = np.transpose(ref["stem"], (0, 2, 3, 1))[:, ::2, ::2]
        assert got["stem.even"].shape == r.shape
        assert np.abs(got["stem.even"] - r).max() / np.abs(r).max() < 3e-3


# A PAFPN level as one launch (csrc/lat_fpn.hip, round 4): lateral 1x1 (+ nearest-2x upsampled coarser lateral) and the 3x3 conv on it; the lateral
# is stored only when a finer level adds it.  Fused and unfused (FID_NO_LATFPN_FUSE=1) lowering against the oracle: two levels (the coarse one's
# lateral is stored AND consumed on chip, the fine one's never leaves the CU), 96- and 64-channel sources, tile-multiple and ragged maps
# (every level exactly twice the next one: the oracle's nearest-2x upsample needs that, as the reference's PAFPN does)
@pytest.mark.parametrize("fuse", [True, False])
@pytest.mark.parametrize("hw,cch,batch", [((80, 80), 88, 2), ((40, 56), 88, 3), ((16, 16), 88, 1), ((52, 36), 64, 2), ((24, 72), 56, 5)])
def test_fused_lateral_fpn(ctx, monkeypatch, fuse, hw, cch, batch):
    from scrfd_arcface_facerecognition_amd import lower
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet
    if not fuse:
        monkeypatch.setenv("FID_NO_LATFPN_FUSE", "1")
    kw = dict(bias=True, post_bn=False)
    net = Net("t", hw, 127.5, 1.0 / 128.0)
    net.add(Conv("s", "input", 3, 32, act="relu"))
    net.add(Conv("c3", "s", 32, cch, act="relu"))
    net.add(Conv("c4", "c3", cch, cch, stride=2, act="relu"))
    net.add(Conv("c5", "c4", cch, cch, stride=2, act="relu"))
    net.add(Conv("lat2", "c5", cch, 56, k=1, pad=0, **kw))
    net.add(Conv("lat1", "c4", cch, 56, k=1, pad=0, res="lat2", res_up2=True, **kw))
    net.add(Conv("lat0", "c3", cch, 56, k=1, pad=0, res="lat1", res_up2=True, **kw))
    net.add(Conv("fpn0", "lat0", 56, 56, **kw))
    net.add(Conv("fpn1", "lat1", 56, 56, **kw))
    net.add(Conv("fpn2", "lat2", 56, 56, **kw))
    net.outputs = ["fpn0", "fpn1", "fpn2"]
    P = archs.synth_params(net, seed=81)
    low = lower.lower(net, P)
    n_fused = sum(int(r[0]) == lower.OP_LATFPN for r in low.ops)
    assert n_fused == (3 if fuse else 0), low.op_names
    if fuse:
        assert "lat0" not in low.tensor_id and "lat1" in low.tensor_id and "lat2" in low.tensor_id     # only the laterals a finer level adds are stored
    images = np.random.default_rng(16).integers(0, 256, (batch,) + hw + (3,), dtype=np.uint8)
    cn = CompiledNet(ctx, net, P, max_batch=batch)
    for _ in range(2):
        cn.run(images)
    got = {o: cn.read(o, batch) for o in net.outputs}
    if fuse:
        got["lat1"] = cn.read("lat1", batch)
    cn.close()
    ref = onets.run_net(net, P, align.blob_from_images(list(images), net.in_scale, net.in_mean), keep=["lat1"])
    for o, g in got.items():
        r = np.transpose(ref[o], (0, 2, 3, 1))
        assert g.shape == r.shape
        assert np.abs(g - r).max() / np.abs(r).max() < 6e-3, (o, hw)

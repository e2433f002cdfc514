"""GPU parity: the fused SCRFD stem (conv/s2 - conv - conv - maxpool in one kernel) against the unfused
layer-by-layer executor and against the fp32 oracle, at frame sizes with full and partial tiles."""
import numpy as np
import pytest

from oracle import align, nets as onets
from scrfd_arcface_facerecognition_amd import archs
from scrfd_arcface_facerecognition_amd.archs import Conv, MaxPool, Net

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from scrfd_arcface_facerecognition_amd._lib import Context
    c = Context(0)
    yield c
    c.close()


def stem_net(hw, c0, c2):
    net = Net("t", hw, 127.5, 1.0 / 128.0)
    net.add(Conv("stem.0", "input", 3, c0, stride=2, act="relu"))
    net.add(Conv("stem.1", "stem.0", c0, c0, act="relu"))
    net.add(Conv("stem.2", "stem.1", c0, c2, act="relu"))
    net.add(MaxPool("stem.pool", "stem.2", c2))
    net.outputs = ["stem.pool"]
    return net


# kernels: the row-structured stem (csrc/stem_rows.hip, the default) with 8 / 6 pooled rows per tile, and the first design on flattened
# fragments (csrc/stem_fused.hip, FID_STEM_OLD=1); "roles" = the row-structured stem with front / back wave roles (FID_STEM_ROLES=1)
@pytest.mark.parametrize("kernel", ["rows8", "rows6", "roles", "flat"])
@pytest.mark.parametrize("hw,c0,c2,batch", [((64, 64), 28, 56, 3), ((96, 160), 28, 56, 2), ((72, 100), 12, 24, 2),
                                             ((320, 320), 28, 56, 1), ((100, 76), 24, 24, 3), ((640, 640), 28, 56, 1)])
def test_fused_stem_matches_oracle_and_unfused(ctx, monkeypatch, kernel, hw, c0, c2, batch):
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet
    from scrfd_arcface_facerecognition_amd.lower import lower
    net = stem_net(hw, c0, c2)
    P = archs.synth_params(net, seed=3)
    images = np.random.default_rng(5).integers(0, 256, (batch,) + hw + (3,), dtype=np.uint8)
    images[0, :5, :7] = 0                      # real zero pixels are NOT padding (they map to -255/256)
    assert lower(net, P).op_names == ["stem.fused"]
    if kernel == "flat":
        monkeypatch.setenv("FID_STEM_OLD", "1")
    elif kernel == "roles":                    # conv0 + conv1 waves one tile ahead of conv2 + pool waves (scrfd_stem_roles)
        monkeypatch.setenv("FID_STEM_ROLES", "1")
    else:
        monkeypatch.setenv("FID_STEM_PY", kernel[4:])
    cn = CompiledNet(ctx, net, P, max_batch=batch)
    cn.run(images)
    fused = cn.read("stem.pool", batch)
    cn.close()
    monkeypatch.setenv("FID_NO_STEM_FUSE", "1")
    assert len(lower(net, P).op_names) == 4
    cn2 = CompiledNet(ctx, net, P, max_batch=batch)
    cn2.run(images)
    unfused = cn2.read("stem.pool", batch)
    cn2.close()
    ref = onets.run_net(net, P, align.blob_from_images(list(images), net.in_scale, net.in_mean))["stem.pool"]
    ref = np.transpose(ref, (0, 2, 3, 1))
    scale = np.abs(ref).max()
    assert np.abs(fused - ref).max() / scale < 6e-3
    assert np.abs(fused - unfused).max() / scale < 6e-3

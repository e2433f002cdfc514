"""CPU: the dependency-free ONNX reader round-trips all five architectures through our own
ONNX writer, with explicit BatchNormalization nodes and with BN folded by the 'exporter'.
No real insightface .onnx exists offline: agreement with the upstream files' node patterns is unpinned."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
from export_onnx import export  # noqa: E402
from oracle import nets as onets  # noqa: E402
from scrfd_arcface_facerecognition_amd import archs  # noqa: E402
from scrfd_arcface_facerecognition_amd.lower import lower  # noqa: E402
from scrfd_arcface_facerecognition_amd.onnx_reader import onnx_to_ir, parse_onnx  # noqa: E402

SMALL = {"scrfd_10g": (64, 64), "scrfd_2.5g": (64, 64), "scrfd_500m": (64, 64), "arcface_r50": (112, 112),
         "arcface_mbf": (112, 112)}


@pytest.mark.parametrize("arch", sorted(SMALL))
def test_round_trip_with_bn_nodes_is_bit_identical(arch):
    net = archs.ARCHS[arch](SMALL[arch])
    P = archs.synth_params(net, seed=4)
    data = export(net, P, fold_bn=False)
    nodes, inits, g_in, g_out, shape = parse_onnx(data)
    assert g_in == ["input.1"] and shape == [1, 3, SMALL[arch][0], SMALL[arch][1]]
    assert len(g_out) == (9 if arch.startswith("scrfd") else 1)
    net2, P2 = onnx_to_ir(data, arch, in_scale=net.in_scale)
    assert [n.kind for n in net2.nodes] == [n.kind for n in net.nodes]
    assert archs.count_macs(net2) == archs.count_macs(net)
    a, b = lower(net, P), lower(net2, P2)
    assert np.array_equal(a.ops[:, :20], b.ops[:, :20]) and np.array_equal(a.tensors, b.tensors)
    assert a.blob == b.blob                      # identical packed weights: the graph + every tensor were recovered


@pytest.mark.parametrize("arch", ["scrfd_2.5g", "arcface_mbf"])
def test_round_trip_with_folded_bn_matches_numerically(arch):
    net = archs.ARCHS[arch](SMALL[arch])
    P = archs.synth_params(net, seed=5)
    net2, P2 = onnx_to_ir(export(net, P, fold_bn=True), arch, in_scale=net.in_scale)
    assert not any(k.endswith(".post_bn.gamma") for k in P2 if k.startswith("conv"))
    x = np.random.default_rng(0).standard_normal((1, 3) + SMALL[arch]).astype(np.float32)
    o1, o2 = onets.run_net(net, P, x), onets.run_net(net2, P2, x)
    for k1, k2 in zip(net.outputs, net2.outputs):
        for u, v in zip(o1[k1] if isinstance(o1[k1], tuple) else (o1[k1],), o2[k2] if isinstance(o2[k2], tuple) else (o2[k2],)):
            assert np.abs(u - v).max() < 1e-4 * max(1.0, np.abs(u).max())


def test_model_path_dispatch(tmp_path):
    from scrfd_arcface_facerecognition_amd.engine import resolve_model
    net = archs.mobilefacenet()
    P = archs.synth_params(net, 0)
    f = tmp_path / "w600k_mbf.onnx"
    f.write_bytes(export(net, P))
    net2, P2 = resolve_model(str(f))
    assert net2.name == "arcface_mbf" and abs(net2.in_scale - 1 / 127.5) < 1e-12 and net2.in_hw == (112, 112)
    assert lower(net2, P2).blob == lower(net, P).blob
    with pytest.raises(FileNotFoundError):
        resolve_model(str(tmp_path / "missing.onnx"))


# graphs that are in no table of archs.py, with node patterns the round trips above do not use: the reader converts by structure
VARIANTS = [
    ("v_det_a", lambda: archs.scrfd_resnet("v_det_a", (96, 96), stem=16, planes=(32, 64, 64, 96), blocks=(1, 2, 1, 2), neck=32, head_ch=64, head_convs=2,
                                           head_shared=False), dict(fold_bn=False, dynamic_reshape=True, upsample="sizes")),
    ("v_det_b", lambda: archs.scrfd_resnet("v_det_b", (96, 96), stem=20, planes=(40, 72, 72, 120), blocks=(2, 1, 2, 1), neck=40, head_ch=48, head_convs=4,
                                           head_shared=True), dict(fold_bn=True, upsample="op9")),
    ("v_mbf_a", lambda: archs.mobilefacenet(blocks=(1, 2, 3, 1)), dict(fold_bn=False, slope_rank=4)),
    ("v_mbf_b", lambda: archs.mobilefacenet(blocks=(3, 1, 2, 2)), dict(fold_bn=True)),
    ("v_ir_a", lambda: archs.iresnet50(layers=(2, 1, 3, 1), name="v_ir_a"), dict(fold_bn=False, slope_rank=4)),
]


@pytest.mark.parametrize("variant", range(len(VARIANTS)))
def test_reader_converts_graphs_outside_the_tables(variant):
    """depths / widths / head layouts of no table; BN unfolded or folded; PRelu slopes [1, C, 1, 1]; Resize by target size and the opset-9 Upsample;
    the head's Reshape fed by Shape -> Gather -> Unsqueeze -> Concat.  The fp32 oracle on the reader's IR equals the oracle on the source graph."""
    from oracle import align as oalign
    base, make, kw = VARIANTS[variant]
    net = make()
    P = archs.synth_params(net, seed=11 + variant)
    rnet, rP = onnx_to_ir(export(net, P, **kw), base)
    assert len(rnet.nodes) == len(net.nodes) and [n.kind for n in rnet.nodes] == [n.kind for n in net.nodes]
    assert abs(rnet.in_scale - net.in_scale) < 1e-12 and rnet.in_hw == net.in_hw
    img = np.random.default_rng(variant).integers(0, 256, (1,) + net.in_hw + (3,), dtype=np.uint8)
    blob = oalign.blob_from_images(list(img), net.in_scale, net.in_mean)
    a, b = onets.run_net(net, P, blob), onets.run_net(rnet, rP, blob)
    for x, y in zip(net.outputs, rnet.outputs):
        for u, v in zip(a[x] if isinstance(a[x], tuple) else (a[x],), b[y] if isinstance(b[y], tuple) else (b[y],)):
            assert np.abs(u - v).max() <= 1e-5 * max(1.0, float(np.abs(u).max()))

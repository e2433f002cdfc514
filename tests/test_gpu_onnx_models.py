"""GPU, SURVEY.md section 8 f-1: `.onnx` files through the reference's constructors.  Each of the five architectures the
reference downloads (download.sh:12-16) is written as an ONNX file (tools/export_onnx.py: our own writer -- no insightface file
exists offline, so the upstream node patterns stay unpinned) under the reference's basename, loaded exactly like
/root/reference/main.py:156-157 does -- `SCRFD(path, input_size=..., conf_thres=...)`, `ArcFace(path)` -- and the HIP results are
compared with the fp32 oracle run on the parameters the file was written from."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
from export_onnx import export  # noqa: E402
from oracle import align as oalign, nets as onets, pipeline as opipe, postprocess as pp  # noqa: E402
from scrfd_arcface_facerecognition_amd import archs  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from scrfd_arcface_facerecognition_amd._lib import default_context
    return default_context(0)


@pytest.mark.parametrize("basename,arch,fold_bn", [("det_10g", "scrfd_10g", False), ("det_2.5g", "scrfd_2.5g", True), ("det_500m", "scrfd_500m", False)])
def test_scrfd_from_onnx_file(ctx, tmp_path, basename, arch, fold_bn):
    from models import SCRFD                                     # the reference's import path (main.py:11)
    from scrfd_arcface_facerecognition_amd.pipeline import calibrate_detector_bias
    rng = np.random.default_rng(31)
    frame = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    det_img, _ = oalign.letterbox(frame)
    net = archs.ARCHS[arch]((640, 640))
    P, _ = calibrate_detector_bias(ctx, net, archs.synth_params(net, seed=8), det_img[None], target=40, max_batch=1)
    path = tmp_path / f"{basename}.onnx"
    path.write_bytes(export(net, P, fold_bn=fold_bn))
    d = SCRFD(str(path), input_size=(640, 640), conf_thres=0.5)                   # main.py:156
    assert d.session.net.name == arch and len(d.output_names) == 9
    heads = d.session.run_images(det_img[None])
    ref = onets.run_net(net, P, oalign.blob_from_images([det_img], net.in_scale, net.in_mean))
    for li, name in enumerate(net.outputs):
        sc, bb, kp = ref[name]
        # (4e-3 at 640x640 like tests/test_gpu_fullsize_properties.py: tools/head_trace.py shows every op adding its fp16 rounding (3-7e-4
        # of the map's maximum) and the stride-32 towers carrying 1.5x the magnitude of the stride-8 ones -- the same relative error is a
        # larger logit error there; no single layer stands out)
        assert np.abs(heads[li] - sc[0]).max() < 4e-3, name
        assert np.abs(heads[3 + li] - bb[0]).max() < 3e-2, name
        assert np.abs(heads[6 + li] - kp[0]).max() < 3e-2, name
    det, kps = d.detect(frame, max_num=0, metric="max")
    odet, okps = pp.detect_from_heads(heads, frame.shape[:2])                       # decisions on the same head tensors: bit-exact
    assert len(odet) > 0 and np.array_equal(det, odet) and np.array_equal(kps, okps)
    # ... and close to what the fp32 net decides: every fp32 detection well clear of the thresholds has a device detection on it
    ref_heads = [ref[n][0][0] for n in net.outputs] + [ref[n][1][0] for n in net.outputs] + [ref[n][2][0] for n in net.outputs]
    fdet, _ = pp.detect_from_heads(ref_heads, frame.shape[:2])
    strong = fdet[fdet[:, 4] > 0.55]
    for r in strong:
        assert (np.abs(det[:, :4] - r[:4]).max(axis=1) < 1.0).any()


@pytest.mark.parametrize("basename,arch,fold_bn", [("w600k_r50", "arcface_r50", False), ("w600k_mbf", "arcface_mbf", True)])
def test_arcface_from_onnx_file(ctx, tmp_path, basename, arch, fold_bn):
    from models import ArcFace
    net = archs.ARCHS[arch]()
    P = archs.synth_params(net, seed=9)
    path = tmp_path / f"{basename}.onnx"
    path.write_bytes(export(net, P, fold_bn=fold_bn))
    r = ArcFace(str(path))                                                          # main.py:157
    assert r.session.net.name == arch and r.input_size == (112, 112)
    rng = np.random.default_rng(32)
    frame = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    kps = np.array([[300, 200], [380, 205], [338, 250], [305, 290], [372, 295]], np.float32)
    emb = r(frame, kps)
    ref, crop = opipe.embed(frame, kps, net, P)
    assert np.array_equal(r.align(frame, kps), crop)
    assert 1 - float(emb @ ref / np.linalg.norm(emb) / np.linalg.norm(ref)) < 1e-3
    assert np.abs(emb / np.linalg.norm(emb) - ref / np.linalg.norm(ref)).max() < 1e-3
    feats = r.get_feat([crop, crop[:, ::-1].copy()])
    # (another batch size picks other kernels / fp32 summation orders: equal to fp16 noise, not bit for bit)
    assert feats.shape == (2, 512) and 1 - float(feats[0] @ emb / np.linalg.norm(feats[0]) / np.linalg.norm(emb)) < 1e-4


# ---- f-1 by STRUCTURE: graphs whose depths, widths and head layouts are in NO table of archs.py, written with node patterns the
# five round-trip files above do not use, loaded through the reference's constructors.  The device result is compared with the fp32 oracle evaluated
# on the READER's IR and parameters (session.net / session.params: what onnx_reader made of the file) AND with the oracle on the graph the file was
# written from -- a reader that mis-converts consistently would pass the first and fail the second.
def _det_variants():
    return [
        ("v_det_a", lambda: archs.scrfd_resnet("v_det_a", (320, 320), stem=16, planes=(32, 64, 64, 96), blocks=(1, 2, 1, 2), neck=32, head_ch=64,
                                               head_convs=2, head_shared=False), dict(fold_bn=False, dynamic_reshape=True, upsample="sizes")),
        ("v_det_b", lambda: archs.scrfd_resnet("v_det_b", (320, 320), stem=20, planes=(40, 72, 72, 120), blocks=(2, 1, 2, 1), neck=40, head_ch=48,
                                               head_convs=4, head_shared=True), dict(fold_bn=True, dynamic_reshape=False, upsample="op9")),
    ]


def _rec_variants():
    return [
        ("v_mbf_a", lambda: archs.mobilefacenet(blocks=(1, 2, 3, 1)), dict(fold_bn=False, slope_rank=4)),
        ("v_mbf_b", lambda: archs.mobilefacenet(blocks=(3, 1, 2, 2)), dict(fold_bn=True, slope_rank=3)),
        ("v_ir_a", lambda: archs.iresnet50(layers=(2, 1, 3, 1), name="v_ir_a"), dict(fold_bn=False, slope_rank=4)),
    ]


def _not_a_table(net):
    """no table of archs.py has this graph's (kind, channels, kernel, stride) sequence"""
    sig = [(n.kind, getattr(n, "cin", 0), getattr(n, "cout", 0), getattr(n, "k", 0), getattr(n, "stride", 0)) for n in net.nodes]
    for f in archs.ARCHS.values():
        t = f(net.in_hw) if net.nodes[-1].kind == "dethead" else f()
        if sig == [(n.kind, getattr(n, "cin", 0), getattr(n, "cout", 0), getattr(n, "k", 0), getattr(n, "stride", 0)) for n in t.nodes]:
            return False
    return True


@pytest.mark.parametrize("variant", range(2))
def test_scrfd_variant_graph_from_onnx_file(ctx, tmp_path, variant):
    from models import SCRFD
    base, make, kw = _det_variants()[variant]
    net = make()
    assert _not_a_table(net)
    P = archs.synth_params(net, seed=21 + variant)
    path = tmp_path / f"{base}.onnx"
    path.write_bytes(export(net, P, **kw))
    d = SCRFD(str(path), input_size=(320, 320), conf_thres=0.5)
    rnet, rP = d.session.net, d.session.params
    assert [n.kind for n in rnet.nodes].count("dethead") == 3 and len(d.output_names) == 9
    img = np.random.default_rng(40 + variant).integers(0, 256, (1, 320, 320, 3), dtype=np.uint8)
    heads = d.session.run_images(img)
    blob = oalign.blob_from_images(list(img), rnet.in_scale, rnet.in_mean)
    for which, (n_, p_) in {"reader": (rnet, rP), "source": (net, P)}.items():
        ref = onets.run_net(n_, p_, blob)
        for li, name in enumerate(n_.outputs):
            sc, bb, kp = ref[name]
            assert np.abs(heads[li] - sc[0]).max() < 3e-3, (which, name)
            assert np.abs(heads[3 + li] - bb[0]).max() < 3e-2, (which, name)
            assert np.abs(heads[6 + li] - kp[0]).max() < 3e-2, (which, name)


@pytest.mark.parametrize("variant", range(3))
def test_arcface_variant_graph_from_onnx_file(ctx, tmp_path, variant):
    from models import ArcFace
    base, make, kw = _rec_variants()[variant]
    net = make()
    assert _not_a_table(net)
    P = archs.synth_params(net, seed=25 + variant)
    path = tmp_path / f"{base}.onnx"
    path.write_bytes(export(net, P, **kw))
    r = ArcFace(str(path))
    rnet, rP = r.session.net, r.session.params
    assert rnet.in_hw == (112, 112) and abs(rnet.in_scale - 1 / 127.5) < 1e-12
    crops = np.random.default_rng(50 + variant).integers(0, 256, (3, 112, 112, 3), dtype=np.uint8)
    feats = r.get_feat(list(crops))
    blob = oalign.blob_from_images(list(crops), rnet.in_scale, rnet.in_mean)
    for which, (n_, p_) in {"reader": (rnet, rP), "source": (net, P)}.items():
        ref = onets.run_net(n_, p_, blob)[n_.outputs[0]].reshape(3, 512)
        for i in range(3):
            assert 1 - float(feats[i] @ ref[i] / np.linalg.norm(feats[i]) / np.linalg.norm(ref[i])) < 1e-3, (which, i)

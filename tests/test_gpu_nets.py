"""GPU parity: the conv-net executor (MFMA implicit-GEMM conv, stem, pooling, depthwise, FC,
BN folding / fusion done by lower.py) against the fp32 torch-CPU oracle on identical weights.
Tolerances: activations are fp16 on the GPU (fp32 accumulate), so intermediate tensors are compared
relative to their own scale; embeddings by cosine (north_star: within 1e-3)."""
import numpy as np
import pytest

from oracle import align, nets as onets
from scrfd_arcface_facerecognition_amd import archs
from scrfd_arcface_facerecognition_amd.archs import Conv, DetHead, FC, MaxPool, Net

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from scrfd_arcface_facerecognition_amd._lib import Context
    c = Context(0)
    yield c
    c.close()


def run_both(ctx, net, P, images, names):
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet
    cn = CompiledNet(ctx, net, P, max_batch=len(images))
    cn.run(images)
    got = {k: cn.read(k, len(images)) for k in names if k in cn.low.tensor_id}      # (a tensor that a fused op keeps on chip has no record)
    blob = align.blob_from_images(list(images), net.in_scale, net.in_mean)
    ref = onets.run_net(net, P, blob, keep=names)
    cn.close()
    return got, ref


def rel_err(got_nhwc, ref_nchw):
    ref = np.transpose(ref_nchw, (0, 2, 3, 1))
    return float(np.abs(got_nhwc - ref).max() / (np.abs(ref).max() + 1e-6))


def small_net(hw, body):
    net = Net("t", hw, 127.5, 1.0 / 128.0)
    body(net)
    return net


CASES = {
    "3x3_s1_c64": lambda n: (n.add(Conv("s", "input", 3, 64, act="relu")), n.add(Conv("c", "s", 64, 64, act="relu"))),
    "3x3_s2_c64_128": lambda n: (n.add(Conv("s", "input", 3, 64, act="prelu")), n.add(Conv("c", "s", 64, 128, stride=2, act="prelu"))),
    "1x1_s2_down": lambda n: (n.add(Conv("s", "input", 3, 64, act="relu")), n.add(Conv("c", "s", 64, 128, k=1, stride=2, pad=0))),
    "odd_channels_28_56_88": lambda n: (n.add(Conv("s", "input", 3, 28, stride=2, act="relu")), n.add(Conv("a", "s", 28, 56, act="relu")),
                                         n.add(Conv("c", "a", 56, 88, stride=2, act="relu"))),
    "avg_down": lambda n: (n.add(Conv("s", "input", 3, 56, act="relu")), n.add(Conv("c", "s", 56, 88, k=1, pad=0, pre_avgpool=True))),
    "prebn_border_residual": lambda n: (n.add(Conv("s", "input", 3, 64, act="prelu")),
                                         n.add(Conv("a", "s", 64, 64, pre_bn=True, act="prelu")),
                                         n.add(Conv("c", "a", 64, 64, res="s"))),
    "fpn_up2_add": lambda n: (n.add(Conv("s0", "input", 3, 64, stride=2, act="relu")), n.add(Conv("s", "s0", 64, 88, k=1, pad=0, act="relu")),
                              n.add(Conv("d", "s", 88, 224, stride=2, act="relu")),
                              n.add(Conv("l2", "d", 224, 56, k=1, pad=0, bias=True, post_bn=False)),
                              n.add(Conv("c", "s", 88, 56, k=1, pad=0, bias=True, post_bn=False, res="l2", res_up2=True))),
    "maxpool": lambda n: (n.add(Conv("s", "input", 3, 56, stride=2, act="relu")), n.add(MaxPool("c", "s", 56))),
    "depthwise": lambda n: (n.add(Conv("s", "input", 3, 128, stride=2, act="prelu")), n.add(Conv("d", "s", 128, 128, groups=128, act="prelu")),
                            n.add(Conv("c", "d", 128, 128, groups=128, stride=2, act="relu"))),
    "wide_256_512": lambda n: (n.add(Conv("s", "input", 3, 64, stride=2, act="relu")), n.add(Conv("a", "s", 64, 256, stride=2, act="relu")),
                               n.add(Conv("c", "a", 256, 512, stride=2, act="prelu"))),
}


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("hw,batch", [((32, 48), 3), ((64, 64), 2)])
def test_layer_cases(ctx, case, hw, batch):
    net = small_net(hw, CASES[case])
    net.outputs = ["c"]
    P = archs.synth_params(net, seed=1)
    rng = np.random.default_rng(7)
    images = rng.integers(0, 256, (batch, hw[0], hw[1], 3), dtype=np.uint8)
    got, ref = run_both(ctx, net, P, images, ["s", "c"])
    if "s" in got:                                     # (3x3_s1_c64: the first conv and the 64 -> 64 conv behind it are one launch, csrc/stem_block.hip)
        assert rel_err(got["s"], ref["s"]) < 3e-3      # stem: exact inputs, fp32 math, fp16 store
    assert rel_err(got["c"], ref["c"]) < 6e-3, case


def test_dethead_and_fc(ctx):
    net = Net("t", (64, 64), 127.5, 1.0 / 128.0)
    net.add(Conv("s", "input", 3, 56, stride=2, act="relu"))
    net.add(Conv("t0", "s", 56, 80, act="relu"))
    net.add(DetHead("h", "t0", 80, 8))
    net.add(Conv("d", "s", 56, 512, stride=2, k=3, act="none"))
    net.add(Conv("e", "d", 512, 512, stride=2, k=3, act="none"))
    net.add(FC("fc", "e", 512, 8, 8, 512))
    net.outputs = ["h", "fc"]
    P = archs.synth_params(net, seed=2)
    images = np.random.default_rng(1).integers(0, 256, (2, 64, 64, 3), dtype=np.uint8)
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet
    cn = CompiledNet(ctx, net, P, max_batch=2)
    cn.run(images)
    fused = cn.read("h", 2)                              # [B,H,W,30]: cls(2) bbox(8) kps(20)
    emb = cn.read("fc", 2).reshape(2, 512)
    cn.close()
    ref = onets.run_net(net, P, align.blob_from_images(list(images), net.in_scale, net.in_mean))
    sc, bb, kp = ref["h"]
    B = 2
    assert np.abs(fused[..., 0:2].reshape(B, -1, 1) - sc).max() < 2e-3
    assert np.abs(fused[..., 2:10].reshape(B, -1, 4) - bb).max() < 2e-2
    assert np.abs(fused[..., 10:30].reshape(B, -1, 10) - kp).max() < 2e-2
    assert np.abs(emb - ref["fc"]).max() / np.abs(ref["fc"]).max() < 5e-3


def cosine(a, b):
    return float((a * b).sum() / (np.linalg.norm(a) * np.linalg.norm(b)))


def test_arcface_r50_embeddings(ctx):
    net = archs.iresnet50()
    P = archs.synth_params(net, seed=0)
    rng = np.random.default_rng(11)
    images = rng.integers(0, 256, (3, 112, 112, 3), dtype=np.uint8)
    got, ref = run_both(ctx, net, P, images, ["fc"])
    e, r = got["fc"].reshape(3, 512), ref["fc"].reshape(3, 512)
    for i in range(3):
        assert 1.0 - cosine(e[i], r[i]) < 1e-3, i
        en, rn = e[i] / np.linalg.norm(e[i]), r[i] / np.linalg.norm(r[i])
        assert np.abs(en - rn).max() < 1e-3, i          # unit-embedding components within 1e-3
    # different faces must stay different (not a degenerate net)
    assert cosine(r[0], r[1]) < 0.999


@pytest.mark.parametrize("table", ["arcface_mbf", "arcface_mbf_small"])
@pytest.mark.parametrize("fusion", ["bottleneck", "dwpw", "none"])
def test_arcface_mbf_embeddings(ctx, monkeypatch, fusion, table):
    """bottleneck (default): each of the 29 (w600k_mbf's size-pinned table, round 5) / 15 (the rounds 1-4 table) blocks as one launch
    (csrc/mbf_block.hip); dwpw: every depthwise layer fused with the pointwise conv behind it (csrc/dwpw.hip; opt-in, FID_DWPW_FUSE=1);
    none: layer by layer"""
    from scrfd_arcface_facerecognition_amd import lower
    monkeypatch.delenv("FID_DWPW_FUSE", raising=False)
    if fusion != "bottleneck":
        monkeypatch.setenv("FID_NO_MBF_FUSE", "1")
    if fusion == "dwpw":
        monkeypatch.setenv("FID_DWPW_FUSE", "1")
    net = archs.ARCHS[table]()
    P = archs.synth_params(net, seed=0)
    kinds = [int(r[0]) for r in lower.lower(net, P).ops]
    nb, ndw = (29, 29) if table == "arcface_mbf" else (15, 16)
    assert (kinds.count(8), kinds.count(7)) == {"bottleneck": (nb, 0), "dwpw": (0, ndw), "none": (0, 0)}[fusion]
    images = np.random.default_rng(12).integers(0, 256, (2, 112, 112, 3), dtype=np.uint8)
    got, ref = run_both(ctx, net, P, images, ["fc"])
    e, r = got["fc"].reshape(2, 512), ref["fc"].reshape(2, 512)
    for i in range(2):
        assert 1.0 - cosine(e[i], r[i]) < 1e-3


@pytest.mark.parametrize("arch", ["scrfd_10g", "scrfd_2.5g", "scrfd_500m"])
def test_scrfd_heads(ctx, arch):
    net = archs.ARCHS[arch]((320, 320))
    P = archs.synth_params(net, seed=0)
    images = np.random.default_rng(13).integers(0, 256, (1, 320, 320, 3), dtype=np.uint8)
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet
    cn = CompiledNet(ctx, net, P, max_batch=1)
    cn.run(images)
    ref = onets.run_net(net, P, align.blob_from_images(list(images), net.in_scale, net.in_mean))
    for name in net.outputs:
        fused = cn.read(name, 1)
        sc, bb, kp = ref[name]
        assert np.abs(fused[..., 0:2].reshape(1, -1, 1) - sc).max() < 3e-3, name       # sigmoid scores
        assert np.abs(fused[..., 2:10].reshape(1, -1, 4) - bb).max() < 3e-2, name      # stride units
        assert np.abs(fused[..., 10:30].reshape(1, -1, 10) - kp).max() < 3e-2, name
    cn.close()


def test_graph_replay_matches_eager(ctx, monkeypatch):
    """FID_GRAPH=1: from the third run on the same frame buffer the launch sequence is replayed as one hipGraph;
    results must be bit-identical to the eager runs, also after new frames are copied into the buffer and after
    another net made the context's split-K scratch grow."""
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet
    monkeypatch.setenv("FID_GRAPH", "1")
    monkeypatch.setenv("FID_AUTOTUNE", "0")               # both nets below on the same (heuristic) plans -> same summation order
    net = archs.scrfd_500m((160, 160))
    P = archs.synth_params(net, seed=3)
    rng = np.random.default_rng(8)
    a, b = (rng.integers(0, 256, (2, 160, 160, 3), dtype=np.uint8) for _ in range(2))
    cn = CompiledNet(ctx, net, P, max_batch=2)
    buf = ctx.to_device(a)
    outs = []
    for i in range(5):
        cn.run_device(buf, 2)
        outs.append([cn.read(k, 2).copy() for k in net.outputs])
    for o in outs[1:]:
        for x, y in zip(o, outs[0]):
            assert np.array_equal(x, y)
    buf.upload(b)
    cn.run_device(buf, 2)                                  # replayed graph, new pixels in the same buffer
    got_b = [cn.read(k, 2).copy() for k in net.outputs]
    monkeypatch.setenv("FID_GRAPH", "0")
    eager = CompiledNet(ctx, net, P, max_batch=2)
    eager.run_device(buf, 2)
    for k, x in zip(net.outputs, got_b):
        assert np.array_equal(x, eager.read(k, 2))
    big = archs.iresnet50()
    other = CompiledNet(ctx, big, archs.synth_params(big, seed=1), max_batch=4)      # grows the shared scratch
    other.run(rng.integers(0, 256, (4, 112, 112, 3), dtype=np.uint8))
    cn.run_device(buf, 2)
    for k, x in zip(net.outputs, got_b):
        assert np.array_equal(x, cn.read(k, 2))
    other.close(); eager.close(); cn.close()


def test_first_run_autotune_at_unseen_shapes(ctx, monkeypatch):
    """Regression for r01's recorded SIGSEGV in fid_net_run (gpurun_out/gpu_tests_13.log: the first net that ran after the
    per-layer autotuner landed crashed on the host -- the plan table `tuned` was indexed before fid_net_create sized it).
    A net whose (input size, batch) the process has never seen runs with the autotuner ON: every conv times its candidates
    on first use (incl. split-K candidates against a small batch-1 workspace), at three batch sizes in a row, and the
    results agree with the heuristic-plan run of the same net."""
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet
    monkeypatch.setenv("FID_AUTOTUNE", "1")
    net = archs.scrfd_500m((224, 288))
    P = archs.synth_params(net, seed=9)
    rng = np.random.default_rng(5)
    images = rng.integers(0, 256, (3, 224, 288, 3), dtype=np.uint8)
    cn = CompiledNet(ctx, net, P, max_batch=3)
    outs = {}
    for b in (1, 3, 2):                                   # each batch size tunes afresh; 1 first (smallest split-K workspace)
        cn.run(images[:b])
        outs[b] = {k: cn.read(k, b) for k in net.outputs}
    cn.close()
    monkeypatch.setenv("FID_AUTOTUNE", "0")
    cn0 = CompiledNet(ctx, net, P, max_batch=3)
    cn0.run(images)
    for k in net.outputs:
        ref = cn0.read(k, 3)
        for b in (1, 2, 3):
            assert np.abs(outs[b][k] - ref[:b]).max() < 2e-2, (k, b)      # other kernels / summation orders: fp16 noise only
    cn0.close()


def test_plan_file_makes_runs_bit_identical(ctx, monkeypatch, tmp_path):
    """FID_PLAN / fid_net_plan_save / fid_net_plan_load: the autotuner's picks are persisted per (device, layer table, op,
    batch); a net that loads them times nothing and reproduces the tuned net's outputs bit for bit (two boxes with the same
    plan file return identical heads), while another layer table ignores the file."""
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet
    net = archs.scrfd_500m((160, 192))
    P = archs.synth_params(net, seed=3)
    images = np.random.default_rng(2).integers(0, 256, (2, 160, 192, 3), dtype=np.uint8)
    plan = tmp_path / "mi355x.plan"
    monkeypatch.setenv("FID_AUTOTUNE", "1")
    monkeypatch.setenv("FID_PLAN", str(plan))
    a = CompiledNet(ctx, net, P, max_batch=2)
    a.run(images)
    ref = {k: a.read(k, 2) for k in net.outputs}
    a.close()
    lines = plan.read_text().splitlines()
    n_conv = sum(1 for l in lines if l.count("|") == 4)
    assert n_conv == len(lines) and n_conv >= 15            # (one line per autotuned conv op; the 1x1 - depthwise - 1x1 chains are fused ops, not tuned)
    monkeypatch.delenv("FID_PLAN")
    monkeypatch.setenv("FID_AUTOTUNE", "0")               # no timing at all: picks come from the file
    b = CompiledNet(ctx, net, P, max_batch=2)
    assert b.load_plan(str(plan)) == n_conv
    b.run(images)
    for k in net.outputs:
        assert np.array_equal(b.read(k, 2), ref[k]), k
    out2 = tmp_path / "copy.plan"
    b.save_plan(str(out2))
    assert sorted(out2.read_text().splitlines()) == sorted(set(lines))
    b.close()
    other = archs.scrfd_500m((160, 160))                  # another layer table: nothing matches
    c = CompiledNet(ctx, other, P, max_batch=2)
    assert c.load_plan(str(plan)) == 0
    c.close()


@pytest.mark.parametrize("n_crops", [500, 585])
def test_arcface_r50_chunk500_vs_oracle(ctx, n_crops):
    """BASELINE configs[3] launches IResNet-50 on chunks of 500 crops (round 5: or 585 = 512 STRIP tiles on the 14x14 stage): the large-batch kernel
    plans (two-tile weights-in-registers convs also on STRIP tiles -- a duplicated crop sits at another lane position of another tile and must still
    give a bit-identical row --, implicit GEMM with the weights in registers on the 7x7 / stride-2 layers) against the fp32 oracle on 4 of the
    crops, plus the size-independent properties on all of them (duplicates give identical rows, every embedding finite)."""
    from scrfd_arcface_facerecognition_amd.engine import CompiledNet
    from oracle import align as oalign, nets as onets
    net = archs.iresnet50()
    P = archs.synth_params(net, 0)
    rng = np.random.default_rng(500)
    crops = rng.integers(0, 256, (n_crops, 112, 112, 3), dtype=np.uint8)
    crops[123] = crops[7]
    cn = CompiledNet(ctx, net, P, max_batch=n_crops)
    cn.run(crops)
    e = cn.read(net.outputs[0], n_crops).reshape(n_crops, -1)
    cn.run(crops)
    assert np.array_equal(e, cn.read(net.outputs[0], n_crops).reshape(n_crops, -1))          # deterministic
    cn.close()
    assert np.isfinite(e).all() and np.array_equal(e[123], e[7])
    for i in (0, 7, 250, n_crops - 1):
        ref = onets.run_net(net, P, oalign.blob_from_images([crops[i]], net.in_scale, net.in_mean))[net.outputs[0]].reshape(-1)
        assert 1 - float(ref @ e[i] / np.linalg.norm(ref) / np.linalg.norm(e[i])) < 1e-3, i
        assert np.abs(ref / np.linalg.norm(ref) - e[i] / np.linalg.norm(e[i])).max() < 1e-3, i


@pytest.mark.parametrize("arch,hw", [("scrfd_10g", (320, 320)), ("scrfd_2.5g", (320, 320)), ("scrfd_500m", (320, 320)),
                                     ("arcface_r50", (112, 112)), ("arcface_mbf", (112, 112))])
def test_truncated_blob_is_refused(ctx, arch, hw, monkeypatch):
    """every blob region a kernel builds a buffer resource from is bounds-checked by fid_net_create (second weight image of the
    shortcut-absorbing convs, both filter banks of the fused residual blocks incl. the two-chunk 64-channel image, the bottleneck's w1 /
    b1 / s1 / depthwise tables, the depthwise + pointwise op's tables, the fused stem's six regions, bias / slope rows of plain convs):
    a blob that ENDS at the start of any region an op names is refused with FID_E_INVALID instead of being read past on the GPU."""
    import ctypes as C
    from scrfd_arcface_facerecognition_amd import _lib
    from scrfd_arcface_facerecognition_amd._lib import FaceIdError, check
    from scrfd_arcface_facerecognition_amd.lower import (OP_BBLOCK, OP_CONV, OP_DWPW, OP_LATFPN, OP_MBBLOCK, OP_STEMBLOCK, OP_STEMFUSED, lower)
    if arch == "scrfd_500m":
        monkeypatch.setenv("FID_DWPW_FUSE", "1")             # so that the table holds OP_DWPW records too
    net = archs.ARCHS[arch](hw)
    low = lower(net, archs.synth_params(net, 0))
    ops = np.ascontiguousarray(low.ops, dtype=np.int32)
    tens = np.ascontiguousarray(low.tensors, dtype=np.int32)

    def create(nbytes):
        h = C.c_void_p()
        check(ctx.lib.fid_net_create(ctx.handle, ops.ctypes.data_as(_lib.c_i32_p), ops.shape[0], tens.ctypes.data_as(_lib.c_i32_p),
                                     tens.shape[0], low.blob, nbytes, net.in_hw[0], net.in_hw[1], 1, C.byref(h)))
        return h
    h = create(len(low.blob))                                # the complete blob is accepted
    check(ctx.lib.fid_net_destroy(ctx.handle, h))
    words = {OP_STEMFUSED: (20, 21, 22, 23, 24, 25), OP_BBLOCK: (20, 21, 22, 23, 24), OP_DWPW: (20, 21, 22), OP_MBBLOCK: (20, 21, 22, 24, 25, 28),
             OP_STEMBLOCK: (20, 21, 22), OP_LATFPN: (20, 21)}      # (round 4's fused ops: W_S_W1 / B1 / S1, W_L_W0 / B0 -- ADVICE r4)
    cuts, kinds = set(), set()
    for op in ops:
        t = int(op[0])
        ws = (13, 15, 16) + words.get(t, ()) + ((29, 30) if t == OP_CONV and op[23] > 0 else ())
        for w in ws:
            if op[w] > 0:
                cuts.add(int(op[w])); kinds.add((t, w))
    assert len(cuts) > 10
    if arch == "arcface_r50":
        assert (OP_CONV, 29) in kinds and (OP_BBLOCK, 22) in kinds
        assert (OP_STEMBLOCK, 20) in kinds and (OP_STEMBLOCK, 21) in kinds
    if arch == "scrfd_10g":
        assert (OP_LATFPN, 20) in kinds and (OP_LATFPN, 21) in kinds
    if arch == "arcface_mbf":
        assert (OP_MBBLOCK, 20) in kinds and (OP_MBBLOCK, 28) in kinds
    if arch == "scrfd_500m":
        assert (OP_DWPW, 20) in kinds
    for cut in sorted(cuts):
        with pytest.raises(FaceIdError):
            create(cut)
    # the two-chunk image of a 64-channel fused block: a blob that holds only its FIRST chunk (the 73 728 bytes round 3 checked) is refused
    for op in ops:
        if int(op[0]) == OP_BBLOCK and int(tens[op[1]][1]) == 64:
            with pytest.raises(FaceIdError):
                create(max(int(op[20]), int(op[22])) + 73728)
            break

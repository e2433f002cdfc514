"""CPU: the build-time ISA check of the hand-counted kernels (tools/check_waitcnt.py) accepts the objects the
committed signatures were derived from and REFUSES a stream that differs -- i.e. the detector detects."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

BUILD = os.path.join(ROOT, "scrfd_arcface_facerecognition_amd", "csrc", "build")
TOOL = os.path.join(ROOT, "tools", "check_waitcnt.py")
SIG = os.path.join(ROOT, "scrfd_arcface_facerecognition_amd", "csrc", "waitcnt.sig")


@pytest.mark.skipif(not os.path.exists(os.path.join(BUILD, "conv_ks.o")), reason="objects not built (python __graft_entry__.py)")
def test_check_passes_on_built_objects_and_fails_on_a_changed_stream(tmp_path):
    objs = [os.path.join(BUILD, f"{u}.o") for u in ("conv_wr", "conv_s2", "conv_gw", "conv_ks")]
    r = subprocess.run([sys.executable, TOOL] + objs, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    # one load fewer in one recorded step (what a compiler that merged two loads would produce): must be reported
    lines = open(SIG).read().splitlines()
    i = next(k for k, l in enumerate(lines) if l.startswith("conv_ks:") and " L3 W" in l)
    lines[i] = lines[i].replace(" L3 W", " L2 W", 1)
    bad = tmp_path / "waitcnt.sig"
    bad.write_text("\n".join(lines) + "\n")
    r = subprocess.run([sys.executable, TOOL, objs[3]], capture_output=True, text=True, env=dict(os.environ, FID_WAITCNT_SIG=str(bad)))
    assert r.returncode == 1 and "CHANGED" in r.stdout, r.stdout + r.stderr


def test_every_handcounted_unit_has_signatures():
    mk = open(os.path.join(ROOT, "scrfd_arcface_facerecognition_amd", "csrc", "Makefile")).read()
    units = next(l for l in mk.splitlines() if l.startswith("HANDCOUNTED")).split("=")[1].split()
    sig = open(SIG).read()
    for u in units:
        assert f"\n{u}:" in sig, u

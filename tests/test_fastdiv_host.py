"""Host check of the multiply-high division used by the persistent conv kernels (csrc/common.h: FastDiv):
q = x / d for every divisor a launch can produce and x up to 2^31 - 1.  Compiled with g++ from the header itself."""
import os
import shutil
import subprocess
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_fastdiv_matches_integer_division(tmp_path):
    src = tmp_path / "fd.cpp"
    src.write_text(textwrap.dedent(f"""
        #include <cstdio>
        #include "{ROOT}/scrfd_arcface_facerecognition_amd/csrc/common.h"
        using namespace fid;
        static unsigned q(unsigned x, const FastDiv &f) {{ return f.d == 1 ? x : (unsigned)(((unsigned long long)x * f.mul) >> 32) >> f.shr; }}
        int main() {{
            long bad = 0;
            for (int d = 1; d < 3000; d++) {{
                const FastDiv f = fastdiv_make(d);
                for (long x = 0; x < 2000000; x += (x < 70000 ? 1 : 997)) bad += q((unsigned)x, f) != (unsigned)(x / d);
            }}
            const int ds[] = {{7, 400, 12800, 65535, 65537, 1000003, 1 << 20, (1 << 30) + 1}};
            for (int d : ds) {{
                const FastDiv f = fastdiv_make(d);
                for (long x = 2147483647L - 50000; x <= 2147483647L; x++) bad += q((unsigned)x, f) != (unsigned)(x / d);
            }}
            printf("%ld\\n", bad);
            return bad != 0;
        }}"""))
    exe = tmp_path / "fd"
    inc = ["-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__"]
    subprocess.run(["g++", "-O2", "-std=c++17", *inc, str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip()
    assert out == "0"

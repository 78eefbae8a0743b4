"""chronoclust_amd/csrc/cc_div.h (one prepared reciprocal for the 2 d quotients of a tentative add) against the compiler's
IEEE division on the GPU, bit for bit: tests/hip/div_exact.hip is compiled with the library's floating-point flags and
run over ~12 M operand pairs."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_shared_denominator_division_is_bit_exact(tmp_path):
    exe = str(tmp_path / "div_exact")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                           os.path.join(HERE, "hip", "div_exact.hip"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    words = out.stdout.split()
    assert int(words[3]) > 10_000_000 and int(words[5]) == 0

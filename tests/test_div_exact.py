"""chronoclust_amd/csrc/cc_div.h (one prepared reciprocal for the 2 d quotients of a tentative add) against the compiler's
IEEE division on the GPU, bit for bit: tests/hip/div_exact.hip is compiled with the library's floating-point flags
(by __graft_entry__.build(), or here if the binary is missing or older than its sources) and run over ~12 M operand pairs."""
import os
import shutil
import subprocess

import pytest

from chronoclust_amd import build as cc_build


def program():
    """The prebuilt binary; rebuilt when hipcc is here and the sources changed; skipped when there is neither."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if os.path.exists(hipcc) or shutil.which("hipcc"):
        return cc_build.build_div_test()
    if os.path.exists(cc_build.DIV_TEST_PROGRAM):
        return cc_build.DIV_TEST_PROGRAM
    pytest.skip("neither hipcc nor a prebuilt tests/hip/_build/div_exact on this box")


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_shared_denominator_division_is_bit_exact():
    exe = program()
    out = subprocess.run([exe], capture_output=True, text=True)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    words = out.stdout.split()
    assert int(words[3]) > 10_000_000 and int(words[5]) == 0

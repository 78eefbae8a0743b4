"""chronoclust_amd/csrc/cc_scan16.h on the GPU: the operand layout of the MFMA prefix test, its accumulation error against the
bound cc_tau16 assumes, and the statement the pruned scan relies on - "a row the test abandons has an exact partial sum beyond
the point's threshold" - over ~5 M (point, row) pairs with thresholds around the true distances (tests/hip/prefix16_check.hip,
compiled with the library's floating-point flags by __graft_entry__.build(), or here if missing or older than its sources)."""
import os
import shutil
import subprocess

import pytest

from chronoclust_amd import build as cc_build


def program():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if os.path.exists(hipcc) or shutil.which("hipcc"):
        return cc_build.build_prefix16_test()
    if os.path.exists(cc_build.P16_TEST_PROGRAM):
        return cc_build.P16_TEST_PROGRAM
    pytest.skip("neither hipcc nor a prebuilt tests/hip/_build/prefix16_check on this box")


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_mfma_prefix_test_is_sound():
    exe = program()
    out = subprocess.run([exe], capture_output=True, text=True)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    w = out.stdout.split()
    pairs, abandoned, violations, layout = int(w[1]), int(w[3]), int(w[5]), int(w[9])
    assert pairs > 4_000_000 and violations == 0 and layout == 0
    assert abandoned > pairs // 4, "the test abandons (almost) nothing: it is not being exercised"
    assert float(w[7]) < 2.0 ** -21

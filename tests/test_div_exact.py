"""chronoclust_amd/csrc/cc_div.h (one prepared reciprocal for the 2 d quotients of a tentative add) against the compiler's
IEEE division on the GPU, bit for bit: tests/hip/div_exact.hip is compiled with the library's floating-point flags
(by __graft_entry__.build(), or here if the binary is missing or older than its sources) and run over ~12 M operand pairs."""
import hashlib
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCE = os.path.join(HERE, "hip", "div_exact.hip")
HEADER = os.path.join(HERE, "..", "chronoclust_amd", "csrc", "cc_div.h")
PROGRAM = os.path.join(HERE, "hip", "_build", "div_exact")


def build_program(force=False):
    """hipcc cross-compiles for gfx950 without a GPU; the binary travels to the GPU box with the tree (git-ignored)."""
    # (keyed to the sources' content, not to file times: a copy of the tree need not keep those)
    digest = hashlib.sha256(open(SOURCE, "rb").read() + open(HEADER, "rb").read()).hexdigest()
    stamp = PROGRAM + ".sha256"
    if not force and os.path.exists(PROGRAM) and os.path.exists(stamp) and open(stamp).read().strip() == digest:
        return PROGRAM
    os.makedirs(os.path.dirname(PROGRAM), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                           SOURCE, "-o", PROGRAM])
    with open(stamp, "w") as f:
        f.write(digest + "\n")
    return PROGRAM


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_shared_denominator_division_is_bit_exact():
    exe = build_program()
    out = subprocess.run([exe], capture_output=True, text=True)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    words = out.stdout.split()
    assert int(words[3]) > 10_000_000 and int(words[5]) == 0

"""Builds the HIP shared library in-tree (hipcc cross-compiles for gfx950 without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libchronoclust_hip.so")
import glob
import hashlib

# the one translation unit first (it includes every header beside it); every file under csrc/ counts for staleness
SOURCES = [os.path.join(_HERE, "csrc", "cc_api.hip")] + sorted(
    p for p in glob.glob(os.path.join(_HERE, "csrc", "*")) if p.endswith((".h", ".hip", ".inc")) and not p.endswith("cc_api.hip"))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "chronoclust_hip.h")
# bit-exactness: no FMA contraction, no fast-math anywhere (host or device)
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math"]


def sources_digest():
    """SHA-256 over the contents of every file the library is compiled from (csrc/*, the C-ABI header) and the compiler
    flags; the build-time defines of a variant are not part of it (a variant is stale when its sources are).  Staleness is
    keyed to this, not to file times: a copied tree (the GPU box receives one) need not keep those, and a fresh-looking
    .so beside changed sources must not be loaded silently (_lib.load refuses it)."""
    hsh = hashlib.sha256()
    for path in SOURCES + [HEADER]:
        hsh.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            hsh.update(f.read())
        hsh.update(b"\0")
    hsh.update(" ".join(FLAGS).encode())
    return hsh.hexdigest()


def is_stale(path=None):
    """True when `path` (default: the in-tree library) was not built from the sources as they are now."""
    path = path or LIB_PATH
    stamp = path + ".sha256"
    if not os.path.exists(path) or not os.path.exists(stamp):
        return True
    with open(stamp) as f:
        return f.read().strip() != sources_digest()


def needs_build():
    return is_stale(LIB_PATH)


def build(force=False, verbose=False, out=None, defines=()):
    """out / defines: a variant of the library under another name with build-time knobs set (kernel experiments,
    loaded through CHRONOCLUST_HIP_LIB)."""
    if out is None and not force and not needs_build():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-D" + x for x in defines] + ["-o", out or LIB_PATH, SOURCES[0]]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open((out or LIB_PATH) + ".sha256", "w") as f:
        f.write(sources_digest() + "\n")
    return out or LIB_PATH


DIV_TEST_SOURCE = os.path.join(os.path.dirname(_HERE), "tests", "hip", "div_exact.hip")
DIV_TEST_PROGRAM = os.path.join(os.path.dirname(_HERE), "tests", "hip", "_build", "div_exact")


def build_div_test(force=False):
    """The device test program of csrc/cc_div.h (tests/hip/div_exact.hip; tests/test_div_exact.py runs it on the GPU box).
    hipcc cross-compiles for gfx950 without a GPU; the binary travels with the tree (git-ignored).  Keyed to the sources'
    content, not to file times: a copy of the tree need not keep those."""
    header = os.path.join(_HERE, "csrc", "cc_div.h")
    digest = hashlib.sha256(open(DIV_TEST_SOURCE, "rb").read() + open(header, "rb").read()).hexdigest()
    stamp = DIV_TEST_PROGRAM + ".sha256"
    if not force and os.path.exists(DIV_TEST_PROGRAM) and os.path.exists(stamp) and open(stamp).read().strip() == digest:
        return DIV_TEST_PROGRAM
    os.makedirs(os.path.dirname(DIV_TEST_PROGRAM), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                           DIV_TEST_SOURCE, "-o", DIV_TEST_PROGRAM])
    with open(stamp, "w") as f:
        f.write(digest + "\n")
    return DIV_TEST_PROGRAM


P16_TEST_SOURCE = os.path.join(os.path.dirname(_HERE), "tests", "hip", "prefix16_check.hip")
P16_TEST_PROGRAM = os.path.join(os.path.dirname(_HERE), "tests", "hip", "_build", "prefix16_check")


def build_prefix16_test(force=False):
    """The device check of csrc/cc_scan16.h (tests/hip/prefix16_check.hip: MFMA operand layout, accumulation error, and the
    statement "abandoned implies beyond the threshold"; tests/test_prefix16.py runs it on the GPU box).  Keyed to the content
    of the kernel sources it includes."""
    hsh = hashlib.sha256(open(P16_TEST_SOURCE, "rb").read())
    for path in SOURCES[1:] + [HEADER]:
        hsh.update(open(path, "rb").read())
    digest = hsh.hexdigest()
    stamp = P16_TEST_PROGRAM + ".sha256"
    if not force and os.path.exists(P16_TEST_PROGRAM) and os.path.exists(stamp) and open(stamp).read().strip() == digest:
        return P16_TEST_PROGRAM
    os.makedirs(os.path.dirname(P16_TEST_PROGRAM), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-w",
                           P16_TEST_SOURCE, "-o", P16_TEST_PROGRAM])
    with open(stamp, "w") as f:
        f.write(digest + "\n")
    return P16_TEST_PROGRAM


if __name__ == "__main__":
    print(build(force=True, verbose=True))

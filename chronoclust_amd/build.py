"""Builds the HIP shared library in-tree (hipcc cross-compiles for gfx950 without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libchronoclust_hip.so")
SOURCES = [os.path.join(_HERE, "csrc", f) for f in ("cc_api.hip", "cc_common.h", "cc_online.h", "cc_scan.h", "cc_validate.h", "cc_seq.h",
                                                   "cc_relaxed.h", "cc_points.h", "cc_offline.h", "cc_comm.h", "cc_csv.h", "cc_policy.h")]
HEADER = os.path.join(os.path.dirname(_HERE), "include", "chronoclust_hip.h")


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(s) > t for s in SOURCES + [HEADER])


def build(force=False, verbose=False, out=None, defines=()):
    """out / defines: a variant of the library under another name with build-time knobs set (kernel experiments,
    loaded through CHRONOCLUST_HIP_LIB)."""
    if out is None and not force and not needs_build():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
           # bit-exactness: no FMA contraction, no fast-math anywhere (host or device)
           "-ffp-contract=off", "-fno-fast-math"] + ["-D" + x for x in defines] + [
           "-o", out or LIB_PATH, SOURCES[0]]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out or LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))

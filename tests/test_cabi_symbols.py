"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol that
include/chronoclust_hip.h declares (no compute calls here: there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

from chronoclust_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "chronoclust_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cc_[a-z_0-9]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    path = build.build()
    lib = ctypes.CDLL(path)
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "library lacks %s" % n
    assert set(names) == set(_lib.SYMBOLS), "ctypes binding and header disagree"


def test_staleness_is_keyed_to_content_not_to_file_times(tmp_path):
    """A library whose stamp does not match the sources beside it is stale however new its file looks, and the loader
    refuses it (the GPU box receives a COPY of the tree with the prebuilt .so: file times say nothing there)."""
    import shutil
    path = build.build()
    assert not build.is_stale(path) and not build.needs_build()
    os.utime(path, (1, 1))  # an ancient file time changes nothing
    assert not build.needs_build()
    copy = str(tmp_path / "libcopy.so")
    shutil.copy(path, copy)
    assert build.is_stale(copy)  # no stamp: unknown provenance
    with open(copy + ".sha256", "w") as f:
        f.write("0" * 64 + "\n")
    assert build.is_stale(copy)  # built from other sources
    old_env, old_lib = os.environ.get("CHRONOCLUST_HIP_LIB"), _lib._lib
    os.environ["CHRONOCLUST_HIP_LIB"] = copy
    _lib._lib = None
    try:
        with pytest.raises(_lib.ChronoclustHipError, match="not built from the sources"):
            _lib.load()
    finally:
        _lib._lib = old_lib
        if old_env is None:
            os.environ.pop("CHRONOCLUST_HIP_LIB", None)
        else:
            os.environ["CHRONOCLUST_HIP_LIB"] = old_env
    shutil.copy(path + ".sha256", copy + ".sha256")
    assert not build.is_stale(copy)


def test_no_silent_cpu_fallback():
    """Without a GPU the product must fail loudly, not compute on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.ChronoclustHipError):
        _lib.Handle(0)


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "chronoclust_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), "%s mentions the oracle" % f


def test_ctypes_structs_match_the_header_layout(tmp_path):
    """Every structure that crosses the boundary, field by field: a C program compiled against include/chronoclust_hip.h
    (plain gcc: the header is C) prints sizeof and the offset of every member; the ctypes mirrors must say the same."""
    import subprocess
    pairs = [("cc_params", _lib.CcParams), ("cc_tuning", _lib.CcTuning), ("cc_stats", _lib.CcStats),
             ("cc_relaxed_stats", _lib.CcRelaxedStats), ("cc_policy_config", _lib.CcPolicyConfig),
             ("cc_policy_carry", _lib.CcPolicyCarry), ("cc_policy_obs", _lib.CcPolicyObs),
             ("cc_policy_decision", _lib.CcPolicyDecision)]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "chronoclust_hip.h"', 'int main(void) {']
    for cname, ctype in pairs:
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for field, _ in ctype._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, field, cname, field))
    lines += ['return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, ctype in pairs:
        assert int(out[cname]) == ctypes.sizeof(ctype), cname
        for field, _ in ctype._fields_:
            assert int(out["%s.%s" % (cname, field)]) == getattr(ctype, field).offset, (cname, field)

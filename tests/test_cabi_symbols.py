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

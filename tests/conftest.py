import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _library_built_from_these_sources():
    """The loader refuses a library that was not built from the sources beside it (chronoclust_amd.build.is_stale: content
    hash, not file times).  Build it once per session when it is stale - hipcc cross-compiles for gfx950 without a GPU;
    on the GPU box the prebuilt library and its stamp travel with the tree and nothing is compiled."""
    from chronoclust_amd import build
    if build.needs_build():
        build.build()
    yield

"""One case of the soak's generators by seed: python tools/one_fuzz.py <forced|fuzz2|fuzz3> <seed> (CHRONOCLUST_HIP_TRACE=1 for the batch trace)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pytest  # noqa: F401,E402
import test_fuzz_parity as F  # noqa: E402
import test_pruned_scan as P  # noqa: E402

kind, seed = sys.argv[1], int(sys.argv[2])
if kind == "forced":
    P.test_forced_pruning_fuzz(seed)
else:
    F.test_fuzz_case(seed, 3 if kind == "fuzz3" else 2)
print("ok", kind, seed)

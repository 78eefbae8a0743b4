"""The scenario of tests/test_pruned_scan.py::test_guessed_thresholds_and_points_whose_outlier_list_starts_with_a_bound with the
library's batch trace (CHRONOCLUST_HIP_TRACE=1): tight populations, then wide ones that appear late."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenarios  # noqa: E402
from chronoclust_amd.clustering.hddstream import HDDStream  # noqa: E402

rng = np.random.default_rng(2025)
n, d, g, late = 600_000, 20, 2000, 6
centres = rng.uniform(0.1, 0.9, (g + late, d))
lab = rng.integers(0, g, n)
for s in range(late):
    start = 2 * n // 3 + s * 12000
    idx = start + np.flatnonzero(rng.random(n - start) < 0.004)
    lab[idx] = g + s
sig = np.where(lab >= g, 0.0225, 0.006)
X = np.ascontiguousarray(np.clip(centres[lab] + rng.normal(0.0, 1.0, (n, d)) * sig[:, None], 0.0, 1.0))
cfg = scenarios.params_to_config(scenarios.blob_params(n))
h = HDDStream(cfg, tuning=dict(window=int(os.environ.get("WIN", "8192"))))
h.online_microcluster_maintenance(X, 0)
print(h.stats())

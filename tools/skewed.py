"""Skewed populations: G blobs of which HB (default three) take the share HEAVY of the events (N points per timepoint, two timepoints
of the same populations; the second one runs on the settled table).  Online-phase time and rate per timepoint, chains
longer than the member list, launches of k_chain_long.  Environment: N, D, G, HEAVY, HB, WIN, LA."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from chronoclust_amd import _lib  # noqa: E402

if __name__ == "__main__":
    n, d, g = int(os.environ.get("N", 2_000_000)), int(os.environ.get("D", 14)), int(os.environ.get("G", 2000))
    heavy = float(os.environ.get("HEAVY", 0.3))
    hb = int(os.environ.get("HB", 3))
    rng = np.random.default_rng(7)
    centres = rng.uniform(0.1, 0.9, (g, d))
    cfg = bench.blob_config(n)
    h = _lib.Handle(0)
    h.set_tuning(window=int(os.environ.get("WIN", "0")), lookahead=int(os.environ.get("LA", "0")), time_kernels=1)
    bench.set_params(h, cfg, n, d)
    for t in range(3):
        lab = rng.integers(hb, g, n)
        big = rng.random(n) < heavy
        lab[big] = rng.integers(0, hb, int(big.sum()))
        X = np.ascontiguousarray(np.clip(centres[lab] + rng.normal(0.0, 0.01, (n, d)), 0.0, 1.0))
        h.points_upload(X)
        h.online_run()
        s = h.stats()
        print("run %d: %.1f ms = %.1f M points/s; rows %d windows %d (lookahead %d) rounds %d truncated %d; long chains %d (laid out %d, replayed %d), k_chain_long launches %d" % (
            t, s["run_ms"], n / s["run_ms"] / 1e3, s["rows"], s["windows"], s["lookahead_windows"], s["rounds"], s["truncated"], s["long_chains"],
            s["long_prepared"], s["long_replayed"], s["long_chain_launches"]), flush=True)

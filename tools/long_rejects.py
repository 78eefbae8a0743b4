"""Long chains with rejected steps: G blobs whose spread sits at the radius threshold, so the chains of a window carry radius
tests that fail.  Prints how many long chains were laid out ahead of k_chain and how many of those had to be replayed
(cc_stats.long_prepared / long_replayed).  Environment: N, D, G, EPS, SIGMAS (comma separated), WIN."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenarios  # noqa: E402
from chronoclust_amd.clustering.hddstream import HDDStream  # noqa: E402

if __name__ == "__main__":
    n, d, g = int(os.environ.get("N", 70000)), int(os.environ.get("D", 6)), int(os.environ.get("G", 5))
    eps = float(os.environ.get("EPS", 0.08))
    for sigma in [float(x) for x in os.environ.get("SIGMAS", "0.015,0.03,0.033,0.036,0.04").split(",")]:
        cfg = scenarios.params_to_config(scenarios.blob_params(n, param_epsilon=eps))
        h = HDDStream(cfg, tuning=dict(window=int(os.environ.get("WIN", 16384))))
        for t in range(2):
            X = scenarios.make_blobs(5100 + t, n, d, g, sigma)
            h.online_microcluster_maintenance(X, t)
        s = h.stats()
        print("sigma %.3f: rows %d windows %d rounds %d | long chains %d prepared %d replayed %d | paths %s" % (
            sigma, s["rows"], s["windows"], s["rounds"], s["long_chains"], s["long_prepared"], s["long_replayed"],
            np.bincount(h.labels_path & 3, minlength=3).tolist()), flush=True)

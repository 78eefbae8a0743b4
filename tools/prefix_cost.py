"""Online time of the first n points of the C2 stream for growing n: what the start-up phase (microclusters being
created) costs compared with the steady state."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from chronoclust_amd import _lib  # noqa: E402

if __name__ == "__main__":
    n, d, g = 1_000_000, 20, 5000
    X = bench.make_blobs(42, n, d, g)
    cfg = bench.blob_config(n)
    h = _lib.Handle(0)
    h.set_tuning(window=int(os.environ.get('WIN', '0')), segments=int(os.environ.get('SEG', '0')), lookahead=int(os.environ.get('LA', '0')))
    prev = 0.0
    prev_m = 0
    for m in [n, 10_000, 20_000, 40_000, 80_000, 160_000, 320_000, 640_000, n]:
        h.reset()
        bench.set_params(h, cfg, n, d)  # mu, omicron as for the whole timepoint
        h.points_upload(X[:m])
        h.online_run()
        s = h.stats()
        print("first %7d points: %.2f ms  (windows %d rounds %d truncated %d rows %d)  increment %.1f ns/point" % (
            m, s["run_ms"], s["windows"], s["rounds"], s["truncated"], s["rows"],
            (s["run_ms"] - prev) * 1e6 / max(1, m - prev_m) if m > prev_m else 0.0), flush=True)
        prev, prev_m = s["run_ms"], m

"""Tuning sweep on the GPU box: one resident C2-shaped workload, several (window, segments, rounds) settings."""
import itertools
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from chronoclust_amd import _lib  # noqa: E402

if __name__ == "__main__":
    n = int(os.environ.get("N", 1_000_000))
    d = int(os.environ.get("D", 20))
    g = int(os.environ.get("G", 5000))
    X = bench.make_blobs(42, n, d, g)
    cfg = bench.blob_config(n)
    h = _lib.Handle(0)
    bench.set_params(h, cfg, n, d)
    h.points_upload(X)
    ref = None
    wins = [int(x) for x in os.environ.get("WINS", "512,1024,2048,4096").split(",")]
    segs = [int(x) for x in os.environ.get("SEGS", "32,64,128").split(",")]
    rnds = [int(x) for x in os.environ.get("RNDS", "2,3").split(",")]
    dsegs = [int(x) for x in os.environ.get("DSEGS", "0").split(",")]
    for win, seg, rd, dseg in itertools.product(wins, segs, rnds, dsegs):
        h.set_tuning(window=win, segments=seg, rounds=rd, time_kernels=int(os.environ.get("TK", "0")), dirty_segments=dseg,
                     lookahead=int(os.environ.get("LA", "0")), early_window=int(os.environ.get("EARLY", "0")))
        best = None
        for rep in range(2):
            h.reset()
            t0 = time.perf_counter()
            h.online_run()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        s = h.stats()
        uid, _ = h.labels_download()
        if ref is None:
            ref = uid
        same = bool(np.array_equal(ref, uid))
        print("win %5d seg %4d dseg %4d rounds %d : %7.1f ms  %6.2f Mpts/s  windows %5d rounds %5d trunc %4d scan_ms %.1f same_labels %s" % (
            win, seg, dseg, rd, best * 1e3, n / best / 1e6, s["windows"], s["rounds"], s["truncated"], s["scan_ms"], same), flush=True)

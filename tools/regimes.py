"""Throughput of the online phase over a grid of table sizes and dimensionalities (synthetic blobs, one cold
timepoint each): a check for performance cliffs outside the benchmark shape."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from chronoclust_amd import _lib  # noqa: E402

if __name__ == "__main__":
    n = int(os.environ.get("N", 500_000))
    for d in (5, 20, 40):
        for g in (50, 500, 5000, 20000):
            X = bench.make_blobs(7, n, d, g)
            cfg = bench.blob_config(n)
            h = _lib.Handle(0)
            bench.set_params(h, cfg, n, d)
            h.points_upload(X)
            h.online_run()
            s = h.stats()
            flops = 4.0 * s["scan_pair_dims"]
            print("d %2d blobs %5d: %7.1f ms  %6.2f Mpts/s | rows %5d windows %4d (lookahead %3d) rounds %4d truncated %3d | scan work %.1f ms at the FP64 rate" % (
                d, g, s["run_ms"], n / s["run_ms"] / 1e3, s["rows"], s["windows"], s["lookahead_windows"], s["rounds"], s["truncated"],
                flops / 39.3e12 * 1e3), flush=True)
            del h

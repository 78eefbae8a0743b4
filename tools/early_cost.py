"""Cost of the start-up phase of the C2 stream (first 160 k points) at several window sizes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from chronoclust_amd import _lib  # noqa: E402

if __name__ == "__main__":
    n, d, g = 1_000_000, 20, 5000
    X = bench.make_blobs(42, n, d, g)
    cfg = bench.blob_config(n)
    for win, early in [(8192, 4096), (8192, 2048), (4096, 4096), (4096, 2048), (2048, 2048), (2048, 1024), (1024, 1024)]:
        for m in (40_000, 160_000):
            h = _lib.Handle(0)
            h.set_tuning(window=win, early_window=early, segments=64)
            best = None
            for rep in range(2):
                h.reset()
                bench.set_params(h, cfg, n, d)
                h.points_upload(X[:m])
                h.online_run()
                s = h.stats()
                best = s["run_ms"] if best is None else min(best, s["run_ms"])
            print("window %5d early %5d first %6d points: %6.2f ms (windows %d rounds %d truncated %d)" % (
                win, early, m, best, s["windows"], s["rounds"], s["truncated"]), flush=True)
            del h

"""Start-up phase of the C2 stream at larger early windows (EARLY list), full window 24576."""
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
    for early in [int(x) for x in os.environ.get("EARLY", "4096,6144,8192,12288,16384").split(",")]:
        for m in (40_000, 160_000, 1_000_000):
            h = _lib.Handle(0)
            h.set_tuning(early_window=early, rounds=int(os.environ.get("ROUNDS", "0")), window=int(os.environ.get("WIN", "0")))
            best = None
            for rep in range(2):
                h.reset()
                bench.set_params(h, cfg, n, d)
                h.points_upload(X[:m])
                h.online_run()
                s = h.stats()
                best = s["run_ms"] if best is None else min(best, s["run_ms"])
            print("early %5d first %7d points: %6.2f ms (windows %d rounds %d truncated %d)" % (
                early, m, best, s["windows"], s["rounds"], s["truncated"]), flush=True)
            del h

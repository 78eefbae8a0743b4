"""One cold timepoint of synthetic blobs: N, D, G from the environment (regime check, see tools/regimes.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from chronoclust_amd import _lib  # noqa: E402

if __name__ == "__main__":
    n, d, g = int(os.environ.get("N", 500000)), int(os.environ.get("D", 20)), int(os.environ.get("G", 50))
    X = bench.make_blobs(7, n, d, g)
    cfg = bench.blob_config(n)
    h = _lib.Handle(0)
    h.set_tuning(lookahead=int(os.environ.get("LA", "0")), window=int(os.environ.get("WIN", "0")))
    bench.set_params(h, cfg, n, d)
    h.points_upload(X)
    for rep in range(int(os.environ.get("REPS", 2))):
        h.reset()
        h.online_run()
        s = h.stats()
        print("d %d blobs %d: %.1f ms %.2f Mpts/s | rows %d windows %d (lookahead %d) rounds %d truncated %d" % (
            d, g, s["run_ms"], n / s["run_ms"] / 1e3, s["rows"], s["windows"], s["lookahead_windows"], s["rounds"], s["truncated"]), flush=True)

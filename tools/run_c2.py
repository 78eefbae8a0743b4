"""One C2 online run with tuning from the environment (WIN, SEG, LA); prints the run statistics."""
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
    h = _lib.Handle(0)
    h.set_tuning(window=int(os.environ.get("WIN", "0")), segments=int(os.environ.get("SEG", "0")),
                 lookahead=int(os.environ.get("LA", "0")), windows_per_sync=int(os.environ.get("WPS", "0")),
                 early_window=int(os.environ.get("EARLY", "0")), dirty_segments=int(os.environ.get("DSEG", "0")))
    bench.set_params(h, cfg, n, d)
    h.points_upload(X)
    h.online_run()
    print(h.stats())

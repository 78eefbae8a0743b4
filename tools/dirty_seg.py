"""Start-up phase (first 160 k points of C2) at several dirty-scan splits (DSEG list)."""
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
    for ds in [int(x) for x in os.environ.get("DSEG", "32,64,128,256").split(",")]:
        h = _lib.Handle(0)
        h.set_tuning(dirty_segments=ds)
        best = None
        for rep in range(3):
            h.reset()
            bench.set_params(h, cfg, n, d)
            h.points_upload(X[:160000])
            h.online_run()
            s = h.stats()
            best = s["run_ms"] if best is None else min(best, s["run_ms"])
        print("dirty_segments %4d: first 160000 points %6.2f ms (windows %d rounds %d)" % (ds, best, s["windows"], s["rounds"]), flush=True)
        del h

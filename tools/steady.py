"""Steady state of the C2 stream in isolation: one full online run builds the table (5 000 pcore MCs), then the same
points are run again without a reset (same daystamp: no decay, every point joins an existing MC).  Prints the
statistics of that second run; with time_kernels the clean scan's launches are timed on their stream.
Environment: WIN, SEG, LA (window, segments, lookahead mode), REPS."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from chronoclust_amd import _lib  # noqa: E402

if __name__ == "__main__":
    n, d, g = int(os.environ.get("N", 1_000_000)), int(os.environ.get("D", 20)), int(os.environ.get("G", 5000))
    X = bench.make_blobs(42, n, d, g)
    cfg = bench.blob_config(n)
    h = _lib.Handle(0)
    h.set_tuning(window=int(os.environ.get("WIN", "0")), segments=int(os.environ.get("SEG", "0")),
                 lookahead=int(os.environ.get("LA", "2")), time_kernels=1,
                 windows_per_sync=int(os.environ.get("WPS", "0")))
    bench.set_params(h, cfg, n, d)
    h.points_upload(X)
    h.online_run()
    print("build-up run:", h.stats(), flush=True)
    for rep in range(int(os.environ.get("REPS", "3"))):
        h.online_run()
        s = h.stats()
        print("steady run %d: %.2f ms = %.1f M points/s; windows %d rounds %d truncated %d; clean scan %d launches, "
              "%.1f us each, %.1f T pair-dims/s inside the scan" % (
                  rep, s["run_ms"], n / s["run_ms"] / 1e3, s["windows"], s["rounds"], s["truncated"],
                  s["scan_launches"], 1e3 * s["scan_ms"] / max(1, s["scan_launches"]),
                  s["scan_pair_dims"] / max(1e-9, s["scan_ms"]) / 1e9), flush=True)
        if s.get("scan_p_launches"):
            print("   pruned scans: %d launches, %d (wave, row) pairs, %.2f %% evaluated in full" % (
                s["scan_p_launches"], s["pruned_scan_rows"], 100.0 * s["pruned_scan_full_rows"] / max(1, s["pruned_scan_rows"])), flush=True)

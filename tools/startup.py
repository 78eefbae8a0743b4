"""The START-UP stretch of the C2 stream in isolation: the first N0 points (default 200 000) of the 1 M x 20 stream with
C2's own thresholds (mu from the full timepoint's N), from an empty table - the windows in which the 5 000 microclusters
are created and promoted, i.e. the validation kernels (dirty scans, k_dseed, k_decide, k_chain, commits) at work.
Prints per repetition the online time; under rocprofv3 (--kernel-trace / --pmc) the kernels of this stretch are what
tools/pmc_validation_summary.py and tools/gaps.py read.  Environment: N0, REPS, LA, WIN, EARLY."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from chronoclust_amd import _lib  # noqa: E402

if __name__ == "__main__":
    n, d, g = 1_000_000, int(os.environ.get("D", 20)), int(os.environ.get("G", 5000))
    n0 = int(os.environ.get("N0", 200_000))
    X = bench.make_blobs(42, n, d, g)[:n0].copy()
    cfg = bench.blob_config(n)
    h = _lib.Handle(0)
    h.set_tuning(window=int(os.environ.get("WIN", "0")), lookahead=int(os.environ.get("LA", "0")), time_kernels=0,
                 early_window=int(os.environ.get("EARLY", "0")), windows_per_sync=int(os.environ.get("WPS", "0")),
                 dirty_segments=int(os.environ.get("DSEG", "0")), segments=int(os.environ.get("SEG", "0")),
                 rounds=int(os.environ.get("ROUNDS", "0")))
    bench.set_params(h, cfg, n, d)
    h.points_upload(X)
    import time
    for rep in range(int(os.environ.get("REPS", "3"))):
        time.sleep(0.02)  # (a gap the trace tools cut the repetitions at)
        h.reset()
        h.online_run()
        s = h.stats()
        print("start-up run %d: %d points in %.2f ms = %.1f M points/s; rows %d windows %d rounds %d truncated %d" % (
            rep, n0, s["run_ms"], n0 / s["run_ms"] / 1e3, s["rows"], s["windows"], s["rounds"], s["truncated"]), flush=True)

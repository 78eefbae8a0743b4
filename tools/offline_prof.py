"""Offline phase alone at the C5 shape (50 000 pcore MCs x 40 dims): builds the table with one online run, then times
cc_offline + export a few times (host wall clock); run under rocprofv3 --kernel-trace --stats for the kernel split."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from chronoclust_amd import _lib  # noqa: E402

if __name__ == "__main__":
    n, d, g = int(os.environ.get("N", 1_000_000)), int(os.environ.get("D", 40)), int(os.environ.get("G", 50_000))
    X = bench.make_blobs(42, n, d, g)
    cfg = bench.blob_config(n, promote_after=4)
    h = _lib.Handle(0)
    bench.set_params(h, cfg, n, d)
    h.points_upload(X)
    h.online_run()
    for rep in range(3):
        t0 = time.perf_counter()
        arrays, _ = h.offline_arrays()
        t1 = time.perf_counter()
        print("offline + export: %.1f ms, pcores %d clusters %d" % (1e3 * (t1 - t0), h.count(_lib.PCORE), len(arrays[2])), flush=True)

"""Time of the offline phase pieces after one C2 online run."""
import os
import sys
import time
import ctypes as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from chronoclust_amd import _lib  # noqa: E402

if __name__ == "__main__":
    n, d, g = 1_000_000, 20, 5000
    X = bench.make_blobs(42, n, d, g)
    cfg = bench.blob_config(n)
    h = _lib.Handle(0)
    bench.set_params(h, cfg, n, d)
    h.points_upload(X)
    for rep in range(3):
        t0 = time.perf_counter()
        h.reset()
        t1 = time.perf_counter()
        h.online_run()
        t2 = time.perf_counter()
        nc = C.c_int32()
        h._check(h._lib.cc_offline(h._h, C.byref(nc), None, None, None, None))
        t3 = time.perf_counter()
        h.offline()
        t4 = time.perf_counter()
        print("reset %.2f ms | online_run %.2f ms (kernel clock %.2f) | cc_offline %.2f ms | Handle.offline (cc_offline + export + dicts) %.2f ms" % (
            (t1 - t0) * 1e3, (t2 - t1) * 1e3, h.stats()["run_ms"], (t3 - t2) * 1e3, (t4 - t3) * 1e3))

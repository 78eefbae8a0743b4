"""Single-GPU runs of the other BASELINE shapes (C4: d = 14, C5: d = 40 with 50 k microclusters) at sizes that
fit the GPU box's host memory: online + offline time and the usual exactness invariants."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from chronoclust_amd import _lib  # noqa: E402

SHAPES = {"C4": (5_000_000, 14, 2000), "C5": (4_000_000, 40, 50_000), "C2": (1_000_000, 20, 5000)}

if __name__ == "__main__":
    for name in sys.argv[1:] or ["C4", "C5"]:
        n, d, g = SHAPES[name]
        X = bench.make_blobs(42, n, d, g)
        cfg = bench.blob_config(n)
        h = _lib.Handle(0)
        h.set_tuning(window=int(os.environ.get("WIN", "0")), segments=int(os.environ.get("SEG", "0")), lookahead=int(os.environ.get("LA", "0")))
        bench.set_params(h, cfg, n, d)
        t0 = time.perf_counter()
        h.points_upload(X)
        t1 = time.perf_counter()
        h.online_run()
        t2 = time.perf_counter()
        clusters, _ = h.offline()
        t3 = time.perf_counter()
        s = h.stats()
        pc = h.export(_lib.PCORE)
        uid, _ = h.labels_download()
        ok = pc["w"].sum() + h.export(_lib.OUTLIER)["w"].sum() == n
        print("%s: N=%d d=%d G=%d | upload %.2fs online %.3fs (%.2f Mpts/s) offline %.3fs | pcore %d clusters %d windows %d "
              "rounds %d trunc %d | weight conserved %s" % (name, n, d, g, t1 - t0, t2 - t1, n / (t2 - t1) / 1e6, t3 - t2,
                                                            len(pc["id"]), len(clusters), s["windows"], s["rounds"], s["truncated"], ok),
              flush=True)
        del h, X

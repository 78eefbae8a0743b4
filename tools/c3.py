"""C3 of BASELINE.json on one GPU: 5 timepoints x 1 M x 20 with drift, decay, tracking by lineage and by
association (the app.run pipeline without CSV I/O).  Prints the time of every phase per timepoint."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenarios  # noqa: E402
from chronoclust_amd.clustering.hddstream import HDDStream  # noqa: E402
from chronoclust_amd.tracking.cluster_tracker import TrackByHistoricalAssociation, TrackByLineage  # noqa: E402

if __name__ == "__main__":
    n = int(os.environ.get("N", 1_000_000))
    g = int(os.environ.get("G", 5000))
    sc = dict(seed=42, n=n, d=20, g=g, sigma=0.01, timepoints=5, drift=0.01, churn=0.02)
    params = scenarios.blob_params(n, param_lambda=0.5)
    cfg = scenarios.params_to_config(params)
    t0 = time.perf_counter()
    Xs = scenarios.make_blob_timepoints(sc, raw=True)
    print("generated %d x %d x %d in %.1f s" % (len(Xs), n, 20, time.perf_counter() - t0), flush=True)
    tuning = {k.lower(): int(v) for k, v in os.environ.items() if k in ("WINDOW", "LOOKAHEAD", "SEGMENTS", "ROUNDS", "EARLY_WINDOW", "WINDOWS_PER_SYNC")}
    h = HDDStream(cfg, tuning=tuning or None)
    lineage, assoc = TrackByLineage(), TrackByHistoricalAssociation(handle=h._h)
    prefetch = os.environ.get("PREFETCH", "1") != "0"
    for t, X in enumerate(Xs):
        a = time.perf_counter()
        h.online_microcluster_maintenance(X, t)
        if prefetch and t + 1 < len(Xs):
            h.prefetch(Xs[t + 1])  # uploaded in the background while this timepoint's records and trackers are built
        b = time.perf_counter()
        for cl in h.cluster_records():
            lineage.add_new_child_cluster(cl)
        c = time.perf_counter()
        lineage.calculate_ids()
        d_ = time.perf_counter()
        assoc.set_current_clusters(lineage.child_clusters)
        assoc.track_cluster_history()
        e = time.perf_counter()
        ids = [cl.id for cl in lineage.child_clusters]
        lineage.transfer_child_to_parent()
        assoc.transfer_current_to_previous()
        s = h.stats()
        print("t=%d: clustering %.3f s (online kernel time %.1f ms, %d windows) | cluster records %.3f s | lineage %.3f s | "
              "association %.3f s | pcores %d outliers %d clusters %d new letters e.g. %s" % (
                  t, b - a, s["run_ms"], s["windows"], c - b, d_ - c, e - d_, len(h.table(0)["id"]), len(h.table(1)["id"]),
                  len(ids), ids[:3]), flush=True)

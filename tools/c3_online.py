"""Online phase only of the C3 stream (5 timepoints x 1 M x 20 with drift, churn and decay): per-timepoint kernel time."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenarios  # noqa: E402
from chronoclust_amd.clustering.hddstream import HDDStream  # noqa: E402

if __name__ == "__main__":
    n = int(os.environ.get("N", 1_000_000))
    sc = dict(seed=42, n=n, d=20, g=5000, sigma=0.01, timepoints=int(os.environ.get("T", 3)), drift=0.01, churn=0.02)
    cfg = scenarios.params_to_config(scenarios.blob_params(n, param_lambda=0.5))
    Xs = scenarios.make_blob_timepoints(sc, raw=True)
    tuning = {k.lower(): int(v) for k, v in os.environ.items() if k in ("WINDOW", "LOOKAHEAD", "SEGMENTS", "ROUNDS")}
    h = HDDStream(cfg, tuning=tuning or None)
    for t, X in enumerate(Xs):
        h.online_microcluster_maintenance(X, t)
        s = h.stats()
        print("t=%d online %.1f ms windows %d (lookahead %d) rounds %d truncated %d rows %d" % (
            t, s["run_ms"], s["windows"], s["lookahead_windows"], s["rounds"], s["truncated"], s["rows"]), flush=True)

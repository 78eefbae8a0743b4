import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, scenarios
from chronoclust_amd.clustering.hddstream import HDDStream
n = int(os.environ.get("N", 2_000_000))
C4 = dict(seed=44, n=n, d=14, g=2000, sigma=0.01, timepoints=7, drift=0.01, churn=0.02)
params = scenarios.blob_params(n, param_lambda=2, param_omicron=0.000004)
cfg = scenarios.params_to_config(params)
Xs = scenarios.make_blob_timepoints(C4, raw=True)
h = HDDStream(cfg)
for t, X in enumerate(Xs):
    if t == int(os.environ.get("TRACE_T", "-1")):
        os.environ["CHRONOCLUST_HIP_TRACE"] = "1"
    h.online_microcluster_maintenance(X, t)
    s = h.stats()
    print("t=%d online %.1f ms %.1f Mpts/s windows %d rounds %d trunc %d la %d seq %d scan_p %d scan_u %d full %.3f pcore %d outlier %d" % (
        t, s["run_ms"], n / s["run_ms"] / 1e3, s["windows"], s["rounds"], s["truncated"], s["lookahead_windows"], s["seq_points"],
        s["scan_p_launches"], s["scan_u_launches"], s["pruned_scan_full_rows"] / max(1, s["pruned_scan_rows"]),
        len(h.table(0)["id"]), len(h.table(1)["id"])), flush=True)

"""Agreement of the RELAXED event-sharded mode with the exact path on data that is not well separated (in-process
groups of 4 / 8 ranks on one GPU): overlapping blobs at d = 5 / 14 / 3.  Prints, per timepoint, the share of points
whose cluster / microcluster matches the exact path's (best one-to-one matching, multi.label_agreement), the table
sizes and the points set aside.  Run on the GPU box from the repo root."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scenarios
from test_relaxed_local import run_relaxed_group
from chronoclust_amd import multi
from chronoclust_amd.clustering.hddstream import HDDStream
for (d, g, sigma, eps, n) in ((5, 60, 0.05, 0.05, 120000), (5, 200, 0.03, 0.05, 120000), (14, 100, 0.06, 0.12, 120000), (3, 12, 0.08, 0.05, 60000)):
    cfg = scenarios.params_to_config(scenarios.blob_params(n, param_epsilon=eps, param_lambda=0.5))
    sc = dict(seed=77, n=n, d=d, g=g, sigma=sigma, timepoints=2, drift=0.005, churn=0.03)
    Xs = scenarios.make_blob_timepoints(sc, raw=True)
    exact = HDDStream(cfg)
    for world, mb in ((4, 4096), (4, 16384), (8, 2048)):
        res = run_relaxed_group(world, Xs, cfg, mb)
        ex = HDDStream(cfg)
        for t, X in enumerate(Xs):
            ex.online_microcluster_maintenance(X, t)
            r = res[0][t]
            print("d %d g %d sigma %.2f | world %d minibatch %d t=%d: by cluster %.4f by microcluster %.4f | MCs %d (exact %d) clusters %d (exact %d) set aside %d" % (
                d, g, sigma, world, mb, t, multi.label_agreement(r["point_cluster"], ex.point_cluster_index()),
                multi.label_agreement(r["labels"], ex.labels_uid), len(r["pcore"]["id"]) + len(r["outlier"]["id"]),
                len(ex.table(0)["id"]) + len(ex.table(1)["id"]), len(r["members"]), len(ex.final_clusters), r["rstats"]["deferred_points"]), flush=True)

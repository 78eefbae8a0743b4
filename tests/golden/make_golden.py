#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/ (TEST INFRASTRUCTURE).

Runs ONLY in the build container: it imports the upstream Python reference
from /root/reference through oracle/ref_harness/refenv.py (no-op numba stand-in
+ left-to-right np.sum proxy), drives it, and writes inputs + expected outputs
as data files.  No reference source text is written anywhere.

    python tests/golden/make_golden.py            # everything
    python tests/golden/make_golden.py c1 blobs   # a subset

Outputs
  c1/           bundled synthetic d0-d4 inputs (data files of the reference's
                integration test), its expected result.csv, the (id, cluster_id)
                columns of the expected cluster_points_D*.csv, and per-timepoint
                dumps of the reference HDDStream state (scaled inputs, per-point
                MC uid, MC tables, clusters in merge order)
  c1_sample_run/  the literal config 1 of BASELINE.json: sample_run_script/sample_run.py's call (no gating file,
                omicron = 4.35e-6) on the same d0-d4 inputs as c1/: result.csv, the five
                cluster_points_D*.csv (gzip) and the per-timepoint state dumps
  nocluster/    the 10-row subset inputs + what the reference writes for them
  tracker_scenarios.json   every scenario of the reference's two tracking
                unit-test files, recorded as call sequences + observed ids
  blob_*.npz    d = 20 / 14 / 40 / 5 / 80 synthetic scenarios (inputs regenerated from
                the seed by tests/scenarios.py) with the reference's per-point
                labels, MC tables, clusters and result.csv text
"""
import gzip
import hashlib
import io
import json
import logging
import os
import shutil
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle", "ref_harness"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refenv  # noqa: E402
import scenarios  # noqa: E402

REF_TESTS = os.path.join(refenv.REFERENCE_ROOT, "chronoclust", "tests")


# --------------------------------------------------------------------------
# instrumentation of the imported reference (no reference file is modified)
# --------------------------------------------------------------------------

class Recorder(object):
    """Records HDDStream state after every online_microcluster_maintenance call."""

    def __init__(self):
        self.calls = []
        self._merge_log = {}

    def install(self):
        from chronoclust.clustering import hddstream as H
        from chronoclust.objects import predecon_mc as P
        rec = self
        self._orig_online = H.HDDStream.online_microcluster_maintenance
        self._orig_merge = P.PredeconMC.merge_mc

        def merge_mc(this, other):
            rec._merge_log.setdefault(id(other), []).append(int(this.id))
            return rec._orig_merge(this, other)

        def online(this, X, daystamp, reset_param=True):
            rec._merge_log = {}
            t0 = time.time()
            rec._orig_online(this, X, daystamp, reset_param)
            rec.calls.append(rec.dump(this, np.asarray(X, dtype=np.float64), daystamp, time.time() - t0))

        P.PredeconMC.merge_mc = merge_mc
        H.HDDStream.online_microcluster_maintenance = online

    def uninstall(self):
        from chronoclust.clustering import hddstream as H
        from chronoclust.objects import predecon_mc as P
        H.HDDStream.online_microcluster_maintenance = self._orig_online
        P.PredeconMC.merge_mc = self._orig_merge

    def dump(self, h, X, daystamp, seconds):
        n, d = X.shape
        out = {"daystamp": daystamp, "X": X, "seconds": seconds,
               "params": np.array([h.pi, h.mu, h.omicron, h.pcore_MC_last_id, h.outlier_MC_last_id],
                                  dtype=np.float64)}
        lab = np.full(n, -1, dtype=np.int64)
        for kind, mcs in (("pcore", h.pcore_MC), ("outlier", h.outlier_MC)):
            m = len(mcs)
            out[kind + "_id"] = np.array([next(iter(mc.id)) for mc in mcs], dtype=np.int64).reshape(m)
            out[kind + "_uid"] = np.array([mc.prev_outlier_id for mc in mcs], dtype=np.int64).reshape(m)
            out[kind + "_w"] = np.array([mc.cumulative_weight for mc in mcs], dtype=np.float64).reshape(m)
            for name, attr in (("cf1", "CF1"), ("cf2", "CF2"), ("cen", "cluster_centroids"),
                               ("pref", "preferred_dimension_vector")):
                out[kind + "_" + name] = np.array([np.asarray(getattr(mc, attr), dtype=np.float64) for mc in mcs],
                                                  dtype=np.float64).reshape(m, d)
            for mc in mcs:
                for idx in mc.points.keys():
                    lab[idx] = mc.prev_outlier_id
        out["labels_uid"] = lab
        cl = h.final_clusters
        out["n_clusters"] = np.array([len(cl)], dtype=np.int64)
        members, offsets, set_order = [], [0], []
        for c in cl:
            order = self._merge_log.get(id(c), [])
            assert set(order) == set(c.id) and len(order) == len(c.id)
            members.extend(order)
            set_order.extend(list(c.id))
            offsets.append(len(members))
        out["cl_members"] = np.array(members, dtype=np.int64)
        out["cl_members_setorder"] = np.array(set_order, dtype=np.int64)
        out["cl_offsets"] = np.array(offsets, dtype=np.int64)
        out["cl_w"] = np.array([c.cumulative_weight for c in cl], dtype=np.float64)
        for name, attr in (("cf1", "CF1"), ("cf2", "CF2"), ("cen", "cluster_centroids"),
                           ("pref", "preferred_dimension_vector")):
            out["cl_" + name] = np.array([np.asarray(getattr(c, attr), dtype=np.float64) for c in cl],
                                         dtype=np.float64).reshape(len(cl), d)
        return out


def save_calls(path, calls, extra=None, keep_x=True):
    flat = {}
    for t, c in enumerate(calls):
        for k, v in c.items():
            if k == "X":
                flat["t%d_xsha" % t] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(v).tobytes()).digest(),
                                                      dtype=np.uint8)
                if not keep_x:
                    continue
            flat["t%d_%s" % (t, k)] = np.asarray(v)
    flat["n_timepoints"] = np.array([len(calls)])
    if extra:
        flat.update(extra)
    np.savez_compressed(path, **flat)


def read_labels(csv_path):
    import pandas as pd
    df = pd.read_csv(csv_path, keep_default_na=False, dtype=str)
    return df["id"].to_numpy().astype(np.int64), df["cluster_id"].to_numpy().astype(str)


def reset_logging():
    root = logging.getLogger()
    for h in list(root.handlers):
        root.removeHandler(h)
        h.close()


# --------------------------------------------------------------------------
# C1: the reference's integration test (normal_test.py:34-51)
# --------------------------------------------------------------------------

def gen_c1():
    from chronoclust import app
    src = os.path.join(REF_TESTS, "integration_test", "test_files")
    dst = os.path.join(HERE, "c1")
    os.makedirs(dst, exist_ok=True)
    for t in range(5):
        shutil.copyfile(os.path.join(src, "dataset", "full_dataset", "synthetic_d%d.csv.gz" % t),
                        os.path.join(dst, "synthetic_d%d.csv.gz" % t))
    shutil.copyfile(os.path.join(src, "dataset", "full_dataset", "gating_centroids.csv"),
                    os.path.join(dst, "gating_centroids.csv"))
    shutil.copyfile(os.path.join(src, "expected_output", "result.csv"), os.path.join(dst, "expected_result.csv"))
    # (id, cluster_id) columns of the reference's committed expected per-point files
    lab = {}
    for t in range(5):
        ids, cl = read_labels(os.path.join(src, "expected_output", "cluster_points_D%d.csv" % t))
        lab["t%d_id" % t], lab["t%d_cluster_id" % t] = ids, cl
    np.savez_compressed(os.path.join(dst, "expected_point_labels.npz"), **lab)

    rec = Recorder()
    rec.install()
    out = tempfile.mkdtemp()
    try:
        data = [os.path.join(dst, "synthetic_d%d.csv.gz" % t) for t in range(5)]
        app.run(data=data, output_directory=out, gating_centroid_file=os.path.join(dst, "gating_centroids.csv"),
                **scenarios.C1_PARAMS)
    finally:
        rec.uninstall()
        reset_logging()
    got = open(os.path.join(out, "result.csv")).read()
    assert got == open(os.path.join(dst, "expected_result.csv")).read(), "reference no longer matches its golden"
    extra = {}
    for t in range(5):
        ids, cl = read_labels(os.path.join(out, "cluster_points_D%d.csv" % t))
        assert (ids == lab["t%d_id" % t]).all() and (cl == lab["t%d_cluster_id" % t]).all()
        # full per-point file text of the run (coordinates included) for the writer parity test
        extra["t%d_points_csv" % t] = np.frombuffer(
            gzip.compress(open(os.path.join(out, "cluster_points_D%d.csv" % t), "rb").read()), dtype=np.uint8)
    save_calls(os.path.join(dst, "hdd_state.npz"), rec.calls, extra)
    shutil.rmtree(out)
    print("c1: ok,", [round(c["seconds"], 2) for c in rec.calls], "s per timepoint in the reference")


# --------------------------------------------------------------------------
# literal config 1: sample_run_script/sample_run.py:5-22 (same inputs as c1, no gating file, omicron 4.35e-6)
# --------------------------------------------------------------------------

def gen_c1_sample_run():
    from chronoclust import app
    src = os.path.join(HERE, "c1")  # synthetic_dataset/ == tests/.../full_dataset/ (verified by cmp below)
    dst = os.path.join(HERE, "c1_sample_run")
    os.makedirs(dst, exist_ok=True)
    data = [os.path.join(src, "synthetic_d%d.csv.gz" % t) for t in range(5)]
    for t in range(5):
        a = gzip.open(os.path.join(refenv.REFERENCE_ROOT, "synthetic_dataset", "synthetic_d%d.csv.gz" % t), "rb").read()
        assert a == gzip.open(data[t], "rb").read(), "synthetic_dataset differs from the integration test's inputs"
    rec = Recorder()
    rec.install()
    out = tempfile.mkdtemp()
    try:
        app.run(data=data, output_directory=out, **scenarios.SAMPLE_RUN_PARAMS)
    finally:
        rec.uninstall()
        reset_logging()
    shutil.copyfile(os.path.join(out, "result.csv"), os.path.join(dst, "expected_result.csv"))
    extra = {}
    for t in range(5):
        ids, cl = read_labels(os.path.join(out, "cluster_points_D%d.csv" % t))
        assert (ids == np.arange(len(ids))).all()
        extra["t%d_cluster_id" % t] = cl
        extra["t%d_points_csv" % t] = np.frombuffer(
            gzip.compress(open(os.path.join(out, "cluster_points_D%d.csv" % t), "rb").read(), mtime=0), dtype=np.uint8)
    save_calls(os.path.join(dst, "hdd_state.npz"), rec.calls, extra, keep_x=False)
    shutil.rmtree(out)
    print("c1_sample_run: pcore/outlier per tp", [(len(c["pcore_id"]), len(c["outlier_id"])) for c in rec.calls])


# --------------------------------------------------------------------------
# no-cluster integration test (no_cluster_test.py:24-43)
# --------------------------------------------------------------------------

def gen_nocluster():
    from chronoclust import app
    src = os.path.join(REF_TESTS, "integration_test", "test_files", "dataset", "subset_dataset")
    dst = os.path.join(HERE, "nocluster")
    os.makedirs(dst, exist_ok=True)
    for t in range(5):
        shutil.copyfile(os.path.join(src, "synthetic_d%d.csv.gz" % t), os.path.join(dst, "synthetic_d%d.csv.gz" % t))
    out = tempfile.mkdtemp()
    rec = Recorder()
    rec.install()
    try:
        app.run(data=[os.path.join(dst, "synthetic_d%d.csv.gz" % t) for t in range(5)], output_directory=out,
                **scenarios.NOCLUSTER_PARAMS)
    finally:
        rec.uninstall()
        reset_logging()
    shutil.copyfile(os.path.join(out, "result.csv"), os.path.join(dst, "expected_result.csv"))
    for t in range(5):
        shutil.copyfile(os.path.join(out, "cluster_points_D%d.csv" % t),
                        os.path.join(dst, "expected_cluster_points_D%d.csv" % t))
    save_calls(os.path.join(dst, "hdd_state.npz"), rec.calls)
    shutil.rmtree(out)
    print("nocluster: ok")


# --------------------------------------------------------------------------
# tracking scenarios recorded from the reference's own unit tests
# --------------------------------------------------------------------------

def gen_tracker():
    import unittest
    import chronoclust.tracking.cluster_tracker as ct
    log = []
    cur = {"test": None, "events": None}

    def cl_desc(c):
        d = {"pcore_ids": list(c.pcore_ids), "weight": None if c.cumulative_weight is None else str(c.cumulative_weight)}
        if c.pcore_objects:
            d["pcores"] = [{"id": sorted(int(x) for x in (p.id if hasattr(p.id, "__iter__") else [p.id])),
                            "centroid": [float(x) for x in p.cluster_centroids],
                            "pref": [float(x) for x in p.preferred_dimension_vector]} for p in c.pcore_objects]
        return d

    o_calc = ct.TrackByLineage.calculate_ids
    o_tr = ct.TrackByLineage.transfer_child_to_parent
    o_track = ct.TrackByHistoricalAssociation.track_cluster_history
    o_tr2 = ct.TrackByHistoricalAssociation.transfer_current_to_previous

    def calc(this):
        before = [cl_desc(c) for c in this.child_clusters]
        o_calc(this)
        cur["events"].append({"op": "lineage", "clusters_in_add_order": before,
                              "ids_after": [c.id for c in this.child_clusters],
                              "pcore_ids_after": [list(c.pcore_ids) for c in this.child_clusters]})

    def tr(this):
        o_tr(this)
        cur["events"].append({"op": "lineage_next"})

    def track(this):
        before = [cl_desc(c) for c in this.current_clusters]
        o_track(this)
        cur["events"].append({"op": "assoc", "clusters": before,
                              "assoc_after": [c.get_historical_associates_as_str() for c in this.current_clusters]})

    def tr2(this):
        o_tr2(this)
        cur["events"].append({"op": "assoc_next"})

    ct.TrackByLineage.calculate_ids = calc
    ct.TrackByLineage.transfer_child_to_parent = tr
    ct.TrackByHistoricalAssociation.track_cluster_history = track
    ct.TrackByHistoricalAssociation.transfer_current_to_previous = tr2
    try:
        loader = unittest.TestLoader()
        for fname in ("unittest_track_by_lineage.py", "unittest_track_by_historical_assoc.py"):
            suite = loader.discover(os.path.join(REF_TESTS, "tracking_test"), pattern=fname,
                                    top_level_dir=refenv.REFERENCE_ROOT)

            def walk(s):
                for t in s:
                    if isinstance(t, unittest.TestSuite):
                        walk(t)
                    else:
                        cur["test"], cur["events"] = t.id(), []
                        res = unittest.TestResult()
                        t.run(res)
                        assert res.wasSuccessful(), (t.id(), res.errors, res.failures)
                        log.append({"test": t.id().split(".")[-1], "events": cur["events"]})
            walk(suite)
    finally:
        ct.TrackByLineage.calculate_ids = o_calc
        ct.TrackByLineage.transfer_child_to_parent = o_tr
        ct.TrackByHistoricalAssociation.track_cluster_history = o_track
        ct.TrackByHistoricalAssociation.transfer_current_to_previous = o_tr2
    with open(os.path.join(HERE, "tracker_scenarios.json"), "w") as f:
        json.dump(log, f, indent=1)
    print("tracker: %d scenarios" % len(log))


# --------------------------------------------------------------------------
# blob scenarios (BASELINE.md section 4 generator), d = 20 / 14 / 40 / 5 / 80
# --------------------------------------------------------------------------

def gen_blobs(names=None):
    from chronoclust import app
    import pandas as pd
    for name, sc in scenarios.BLOB_SCENARIOS.items():
        if names and name not in names:
            continue
        Xs = scenarios.make_blob_timepoints(sc, raw=True)
        tmp = tempfile.mkdtemp()
        files = []
        cols = ["m%d" % i for i in range(sc["d"])]
        for t, X in enumerate(Xs):
            fn = os.path.join(tmp, "tp%d.csv" % t)
            pd.DataFrame(X, columns=cols).to_csv(fn, index=False)  # repr floats: exact round trip
            assert (pd.read_csv(fn).to_numpy() == scenarios.through_csv(X)).all()
            files.append(fn)
        out = os.path.join(tmp, "out")
        os.makedirs(out)
        rec = Recorder()
        rec.install()
        try:
            app.run(data=files, output_directory=out, normalise_data=sc.get("normalise", False), **sc["params"])
        finally:
            rec.uninstall()
            reset_logging()
        extra = {"result_csv": np.frombuffer(open(os.path.join(out, "result.csv"), "rb").read(), dtype=np.uint8)}
        for t in range(len(Xs)):
            ids, cl = read_labels(os.path.join(out, "cluster_points_D%d.csv" % t))
            assert (ids == np.arange(len(ids))).all()
            extra["t%d_cluster_id" % t] = cl
        save_calls(os.path.join(HERE, "blob_%s.npz" % name), rec.calls, extra, keep_x=sc.get("normalise", False))
        shutil.rmtree(tmp)
        print("blob %s: pcore/outlier per tp %s, %s s" % (
            name, [(len(c["pcore_id"]), len(c["outlier_id"]), int(c["n_clusters"][0])) for c in rec.calls],
            [round(c["seconds"], 1) for c in rec.calls]))


if __name__ == "__main__":
    refenv.load()
    what = sys.argv[1:] or ["c1", "c1_sample_run", "nocluster", "tracker", "blobs"]
    if "c1" in what:
        gen_c1()
    if "c1_sample_run" in what:
        gen_c1_sample_run()
    if "nocluster" in what:
        gen_nocluster()
    if "tracker" in what:
        gen_tracker()
    if "blobs" in what:
        gen_blobs()
    for w in what:
        if w.startswith("blob:"):
            gen_blobs([w[5:]])

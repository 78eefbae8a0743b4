"""GPU end-to-end tests of chronoclust_amd.app.run against the reference's committed / recorded outputs."""
import gzip
import json
import os

import numpy as np
import pandas as pd
import pytest

import scenarios
from golden_util import GOLDEN

pytestmark = pytest.mark.gpu


def _reset_logging():
    import logging
    root = logging.getLogger()
    for h in list(root.handlers):
        root.removeHandler(h)
        h.close()


def _labels(csv_path):
    df = pd.read_csv(csv_path, keep_default_na=False, dtype=str)
    return df["id"].to_numpy().astype(np.int64), df["cluster_id"].to_numpy().astype(str)


def test_c1_integration_golden(tmp_path):
    """chronoclust/tests/integration_test/normal_test.py: result.csv byte for byte, every (id, cluster_id),
    and the whole per-point files of the recorded reference run."""
    from chronoclust_amd import app
    c1 = os.path.join(GOLDEN, "c1")
    data = [os.path.join(c1, "synthetic_d%d.csv.gz" % t) for t in range(5)]
    try:
        app.run(data=data, output_directory=str(tmp_path), gating_centroid_file=os.path.join(c1, "gating_centroids.csv"),
                **scenarios.C1_PARAMS)
    finally:
        _reset_logging()
    with open(os.path.join(c1, "expected_result.csv"), newline="") as f:
        exp = f.read()
    with open(os.path.join(str(tmp_path), "result.csv"), newline="") as f:
        got = f.read()
    assert got.replace("\r\n", "\n") == exp.replace("\r\n", "\n")
    lab = np.load(os.path.join(c1, "expected_point_labels.npz"))
    rec = np.load(os.path.join(c1, "hdd_state.npz"))
    for t in range(5):
        ids, cl = _labels(os.path.join(str(tmp_path), "cluster_points_D%d.csv" % t))
        np.testing.assert_array_equal(ids, lab["t%d_id" % t])
        assert (cl == lab["t%d_cluster_id" % t]).all()
        exp_text = gzip.decompress(rec["t%d_points_csv" % t].tobytes())
        assert open(os.path.join(str(tmp_path), "cluster_points_D%d.csv" % t), "rb").read() == exp_text
    assert os.path.exists(os.path.join(str(tmp_path), "parameters.csv"))
    # logs/Chronoclust.log: logging.basicConfig is a no-op under pytest (root handlers exist), as upstream


def test_sample_run_script_verbatim(tmp_path, monkeypatch):
    """BASELINE.json config 1: the reference's sample_run_script/sample_run.py:1-22, statement for statement - the
    `chronoclust` import path, the bundled d0-d4 files addressed relative to the working directory, its config dict
    (omicron 4.35e-6, no gating file) and its call - against the outputs of the reference itself."""
    import shutil
    c1 = os.path.join(GOLDEN, "c1")
    os.makedirs(os.path.join(str(tmp_path), "synthetic_dataset"))
    os.makedirs(os.path.join(str(tmp_path), "sample_run_script", "output"))
    for t in range(5):
        shutil.copyfile(os.path.join(c1, "synthetic_d%d.csv.gz" % t),
                        os.path.join(str(tmp_path), "synthetic_dataset", "synthetic_d%d.csv.gz" % t))
    monkeypatch.chdir(os.path.join(str(tmp_path), "sample_run_script"))
    try:
        from chronoclust import app

        data_directory = '../synthetic_dataset'
        data_files = ['{}/synthetic_d{}.csv.gz'.format(data_directory, x) for x in range(5)]
        config = {"beta": 0.2, "delta": 0.05, "epsilon": 0.03, "lambda": 2, "k": 4, "mu": 0.01, "pi": 3,
                  "omicron": 0.00000435, "upsilon": 6.5}
        output_directory = 'output'
        app.run(data=data_files, output_directory=output_directory, param_beta=config['beta'],
                param_delta=config['delta'], param_epsilon=config['epsilon'], param_lambda=config['lambda'],
                param_k=config['k'], param_mu=config['mu'], param_pi=config['pi'], param_omicron=config['omicron'],
                param_upsilon=config['upsilon'])
    finally:
        _reset_logging()
    import chronoclust_amd.app
    assert app is chronoclust_amd.app
    exp_dir = os.path.join(GOLDEN, "c1_sample_run")
    assert open(os.path.join("output", "result.csv"), "rb").read() == \
        open(os.path.join(exp_dir, "expected_result.csv"), "rb").read()
    rec = np.load(os.path.join(exp_dir, "hdd_state.npz"))
    for t in range(5):
        assert open(os.path.join("output", "cluster_points_D%d.csv" % t), "rb").read() == \
            gzip.decompress(rec["t%d_points_csv" % t].tobytes())
    with open(os.path.join("output", "parameters.csv")) as f:
        assert f.read().splitlines() == ["beta,delta,epsilon,lambda,k,mu,pi,omicron,upsilon",
                                         "0.2,0.05,0.03,2,4,0.01,3,4.35e-06,6.5"]


def test_nocluster_integration_golden(tmp_path):
    """chronoclust/tests/integration_test/no_cluster_test.py: no result rows, every point labelled None."""
    from chronoclust_amd import app
    nc = os.path.join(GOLDEN, "nocluster")
    try:
        app.run(data=[os.path.join(nc, "synthetic_d%d.csv.gz" % t) for t in range(5)], output_directory=str(tmp_path),
                **scenarios.NOCLUSTER_PARAMS)
    finally:
        _reset_logging()
    assert open(os.path.join(str(tmp_path), "result.csv"), "rb").read() == \
        open(os.path.join(nc, "expected_result.csv"), "rb").read()
    for t in range(5):
        assert open(os.path.join(str(tmp_path), "cluster_points_D%d.csv" % t), "rb").read() == \
            open(os.path.join(nc, "expected_cluster_points_D%d.csv" % t), "rb").read()
        ids, cl = _labels(os.path.join(str(tmp_path), "cluster_points_D%d.csv" % t))
        assert len(ids) == 10 and (cl == "None").all()


@pytest.mark.parametrize("name", sorted(scenarios.BLOB_SCENARIOS))
def test_blob_end_to_end(name, tmp_path):
    """d = 20 / 14 / 40 / 5 / 80 scenarios through app.run: result.csv bytes (lineage + association strings, pcore id
    set order, rounded weights and centroids) and per-point cluster ids of the recorded reference run."""
    from chronoclust_amd import app
    sc = scenarios.BLOB_SCENARIOS[name]
    z = np.load(os.path.join(GOLDEN, "blob_%s.npz" % name))
    Xs = scenarios.make_blob_timepoints(sc, raw=True)
    files = []
    for t, X in enumerate(Xs):
        fn = os.path.join(str(tmp_path), "tp%d.csv" % t)
        pd.DataFrame(X, columns=["m%d" % i for i in range(sc["d"])]).to_csv(fn, index=False)
        files.append(fn)
    out = os.path.join(str(tmp_path), "out")
    os.makedirs(out)
    try:
        app.run(data=files, output_directory=out, normalise_data=sc.get("normalise", False), **sc["params"])
    finally:
        _reset_logging()
    assert open(os.path.join(out, "result.csv"), "rb").read() == z["result_csv"].tobytes()
    for t in range(len(Xs)):
        ids, cl = _labels(os.path.join(out, "cluster_points_D%d.csv" % t))
        assert (ids == np.arange(len(ids))).all()
        assert (cl == z["t%d_cluster_id" % t]).all()


def _assoc_scenarios():
    with open(os.path.join(GOLDEN, "tracker_scenarios.json")) as f:
        data = json.load(f)
    return [s for s in data if any(e["op"] == "assoc" for e in s["events"])]


@pytest.mark.parametrize("scenario", _assoc_scenarios(), ids=lambda s: s["test"])
def test_association_scenarios_from_reference_unit_tests(scenario):
    """chronoclust/tests/tracking_test/unittest_track_by_historical_assoc.py replayed through the K9 kernel."""
    from chronoclust_amd.objects.cluster import Cluster, PcoreSnapshot
    from chronoclust_amd.tracking.cluster_tracker import TrackByHistoricalAssociation, TrackByLineage
    lineage, assoc = TrackByLineage(), TrackByHistoricalAssociation()
    for ev in scenario["events"]:
        if ev["op"] == "lineage":
            for c in ev["clusters_in_add_order"]:
                cl = Cluster(list(c["pcore_ids"]))
                for p in c.get("pcores", []):
                    cl.pcore_objects.append(PcoreSnapshot(p["id"][0], np.array(p["centroid"]), np.array(p["pref"])))
                lineage.add_new_child_cluster(cl)
            lineage.calculate_ids()
            assert [c.id for c in lineage.child_clusters] == ev["ids_after"]
        elif ev["op"] == "lineage_next":
            lineage.transfer_child_to_parent()
        elif ev["op"] == "assoc":
            if lineage.child_clusters:
                assoc.set_current_clusters(lineage.child_clusters)
            else:  # the first reference test drives the tracker without a lineage pass
                cls = []
                for c in ev["clusters"]:
                    cl = Cluster(list(c["pcore_ids"]))
                    for p in c.get("pcores", []):
                        cl.pcore_objects.append(PcoreSnapshot(p["id"][0], np.array(p["centroid"]), np.array(p["pref"])))
                    cls.append(cl)
                assoc.set_current_clusters(cls)
            assoc.track_cluster_history()
            assert [c.get_historical_associates_as_str() for c in assoc.current_clusters] == ev["assoc_after"]
        elif ev["op"] == "assoc_next":
            assoc.transfer_current_to_previous()


def test_restore_program_continues_exactly(tmp_path):
    """Stop after three timepoints, restart with restore_program=True on all five files: the outputs are the
    reference's golden again (the reference's own resume cannot do this)."""
    from chronoclust_amd import app
    c1 = os.path.join(GOLDEN, "c1")
    data = [os.path.join(c1, "synthetic_d%d.csv.gz" % t) for t in range(5)]
    gating = os.path.join(c1, "gating_centroids.csv")
    out = str(tmp_path)
    try:
        # the scaler is fitted on all five files in both runs, as a user resuming the same job would have it
        import chronoclust_amd.app as A
        real_enumerate = enumerate

        class Stop(Exception):
            pass

        def run_first_three():
            orig = A.save_program_state

            def save_and_maybe_stop(h, o, ta, tl):
                orig(h, o, ta, tl)
                if h.last_data_timestamp == 2:
                    raise Stop()
            A.save_program_state = save_and_maybe_stop
            try:
                app.run(data=data, output_directory=out, gating_centroid_file=gating, **scenarios.C1_PARAMS)
            except Stop:
                pass
            finally:
                A.save_program_state = orig
        run_first_three()
        _reset_logging()
        assert open(os.path.join(out, "result.csv")).read().count("\n") < 24
        app.run(data=data, output_directory=out, gating_centroid_file=gating, restore_program=True,
                **scenarios.C1_PARAMS)
    finally:
        _reset_logging()
    with open(os.path.join(c1, "expected_result.csv"), newline="") as f:
        exp = f.read()
    with open(os.path.join(out, "result.csv"), newline="") as f:
        got = f.read()
    assert got.replace("\r\n", "\n") == exp.replace("\r\n", "\n")
    rec = np.load(os.path.join(c1, "hdd_state.npz"))
    for t in range(5):
        exp_text = gzip.decompress(rec["t%d_points_csv" % t].tobytes())
        assert open(os.path.join(out, "cluster_points_D%d.csv" % t), "rb").read() == exp_text


def test_restore_after_crash_between_result_rows_and_image(tmp_path):
    """A crash after timepoint 3's rows were appended to result.csv but before its image was saved: the resumed run
    re-processes timepoint 3 from the image of timepoint 2 and must not leave its rows in result.csv twice."""
    from chronoclust_amd import app
    import chronoclust_amd.app as A
    c1 = os.path.join(GOLDEN, "c1")
    data = [os.path.join(c1, "synthetic_d%d.csv.gz" % t) for t in range(5)]
    gating = os.path.join(c1, "gating_centroids.csv")
    out = str(tmp_path)

    class Crash(Exception):
        pass

    orig = A.save_program_state

    def crash_before_saving_t3(h, o, ta, tl):
        if h.last_data_timestamp == 3:
            raise Crash()
        orig(h, o, ta, tl)
    try:
        A.save_program_state = crash_before_saving_t3
        try:
            app.run(data=data, output_directory=out, gating_centroid_file=gating, **scenarios.C1_PARAMS)
        except Crash:
            pass
        finally:
            A.save_program_state = orig
        _reset_logging()
        rows_before = open(os.path.join(out, "result.csv")).read()
        assert "\n3," in rows_before  # timepoint 3's rows are there, its image is not
        assert not os.path.exists(os.path.join(out, "program_images", "hddstream.npz.tmp"))
        app.run(data=data, output_directory=out, gating_centroid_file=gating, restore_program=True,
                **scenarios.C1_PARAMS)
    finally:
        _reset_logging()
    with open(os.path.join(c1, "expected_result.csv"), newline="") as f:
        exp = f.read()
    with open(os.path.join(out, "result.csv"), newline="") as f:
        got = f.read()
    assert got.replace("\r\n", "\n") == exp.replace("\r\n", "\n")


@pytest.mark.parametrize("name", ["d5_norm", "d20"])
def test_binary_side_input_gives_the_same_files(name, tmp_path):
    """`.npy` timepoints (no text parse) against the same values read from CSV: result.csv and the
    per-point files must be byte-identical (d5_norm runs with normalise_data=True: device-side scaler)."""
    from chronoclust_amd import app
    sc = scenarios.BLOB_SCENARIOS[name]
    Xs = scenarios.make_blob_timepoints(sc)  # what pd.read_csv returns for the scenario's CSV files
    outs = []
    for kind in ("csv", "npy"):
        files = []
        for t, X in enumerate(Xs):
            fn = os.path.join(str(tmp_path), "tp%d.%s" % (t, kind))
            if kind == "csv":
                pd.DataFrame(scenarios.make_blob_timepoints(sc, raw=True)[t],
                             columns=["m%d" % i for i in range(sc["d"])]).to_csv(fn, index=False)
            else:
                np.save(fn, X)
            files.append(fn)
        out = os.path.join(str(tmp_path), "out_" + kind)
        os.makedirs(out)
        try:
            app.run(data=files, output_directory=out, normalise_data=sc.get("normalise", False), **sc["params"])
        finally:
            _reset_logging()
        outs.append(out)
    for fn in ["result.csv"] + ["cluster_points_D%d.csv" % t for t in range(len(Xs))]:
        assert open(os.path.join(outs[0], fn), "rb").read() == open(os.path.join(outs[1], fn), "rb").read(), fn

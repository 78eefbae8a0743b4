"""CPU tests of the host-side logic that mirrors the reference: scaler arithmetic, lineage strings."""
import json
import os

import numpy as np
import pytest

from golden_util import GOLDEN
from chronoclust_amd.objects.cluster import Cluster
from chronoclust_amd.scaling.scaler import Scaler
from chronoclust_amd.tracking.cluster_tracker import TrackByLineage


def test_scaler_is_bit_identical_to_sklearn_minmax():
    from sklearn.preprocessing import MinMaxScaler
    rng = np.random.default_rng(0)
    for shape, scale in (((500, 3), 40.0), ((2000, 20), 1.0), ((50, 7), 1e6)):
        X = rng.normal(0, 1, shape) * scale + rng.uniform(-5, 5, shape[1])
        X[:, 0] = 3.25  # constant column: range 0 -> scale 1
        ref = MinMaxScaler().fit(X)
        s = Scaler()
        s.fit_scaler(X)
        Y = rng.normal(0, 1, (100, shape[1])) * scale
        assert np.array_equal(s.scale_data(Y), ref.transform(Y))
        Z = ref.transform(Y)
        assert np.array_equal(s.reverse_scaling(Z), ref.inverse_transform(Z))
        assert np.array_equal(s.reverse_scaling([Z[0]]), ref.inverse_transform([Z[0]]))


def _scenarios(op):
    with open(os.path.join(GOLDEN, "tracker_scenarios.json")) as f:
        data = json.load(f)
    return [s for s in data if any(e["op"] == op for e in s["events"])]


@pytest.mark.parametrize("scenario", _scenarios("lineage"), ids=lambda s: s["test"])
def test_lineage_scenarios_from_reference_unit_tests(scenario):
    """Every scenario of chronoclust/tests/tracking_test/*.py, replayed: same cluster ids in the same order."""
    from decimal import Decimal
    tracker = TrackByLineage()
    for ev in scenario["events"]:
        if ev["op"] == "lineage":
            for c in ev["clusters_in_add_order"]:
                w = None if c["weight"] is None else Decimal(c["weight"])
                tracker.add_new_child_cluster(Cluster(list(c["pcore_ids"]), cumulative_weight=w))
            tracker.calculate_ids()
            assert [c.id for c in tracker.child_clusters] == ev["ids_after"]
            assert [list(c.pcore_ids) for c in tracker.child_clusters] == ev["pcore_ids_after"]
        elif ev["op"] == "lineage_next":
            tracker.transfer_child_to_parent()


def test_letters_run_past_z():
    t = TrackByLineage()
    got = [t.get_new_letter() for _ in range(26 * 2 + 2)]
    assert got[25:28] == ["Z", "AA", "BB"] and got[51:54] == ["ZZ", "AAA", "BBB"]


def test_binary_side_input_readers(tmp_path):
    """`.npy` timepoints: values as stored, column names from `<file>.columns` or m0.. by default;
    CSV files keep the reference's reader (header row, pd.read_csv)."""
    import numpy as np
    import pandas as pd
    from chronoclust_amd import app
    from chronoclust_amd.scaling.scaler import read_timepoint
    X = np.random.default_rng(0).random((7, 3))
    npy = str(tmp_path / "tp0.npy")
    np.save(npy, X)
    assert np.array_equal(read_timepoint(npy), X)
    assert app.get_dataset_attributes(npy) == ["m0", "m1", "m2"]
    with open(npy + ".columns", "w") as f:
        f.write("CD4\nCD8\nLy6C\n")
    assert app.get_dataset_attributes(npy) == ["CD4", "CD8", "Ly6C"]
    csv = str(tmp_path / "tp0.csv")
    pd.DataFrame(X, columns=["a", "b", "c"]).to_csv(csv, index=False)
    assert app.get_dataset_attributes(csv) == ["a", "b", "c"]
    assert np.allclose(read_timepoint(csv), X, rtol=0, atol=1e-15)


def test_drop_rows_after_keeps_header_and_earlier_timepoints(tmp_path):
    """restore_program: rows of timepoints the saved image does not cover are removed before appending again
    (chronoclust_amd/app.py:drop_rows_after); quoting of lineage ids with commas survives the rewrite."""
    import csv
    from chronoclust_amd.app import append_to_file, drop_rows_after, write_file_header
    fn = str(tmp_path / "result.csv")
    write_file_header(fn, ["timepoint", "cumulative_size", "tracking_by_lineage"])
    append_to_file(fn, [[0, "10.0", "A"], [1, "12.5", "(A,B)"], [2, "3.0", "(A,B)|1"], [2, "4.0", "C"]])
    before = open(fn, newline="").read()
    drop_rows_after(fn, 2)
    assert open(fn, newline="").read() == before
    drop_rows_after(fn, 1)
    with open(fn, newline="") as f:
        rows = list(csv.reader(f))
    assert rows == [["timepoint", "cumulative_size", "tracking_by_lineage"], ["0", "10.0", "A"], ["1", "12.5", "(A,B)"]]
    assert open(fn, newline="").read() == before[:before.index("2,3.0")]


def test_drop_rows_after_drops_a_torn_last_row(tmp_path):
    """A crash in the middle of append_to_file leaves a short last row (or one whose timepoint is cut): it is dropped
    like the rows of uncovered timepoints instead of blocking the resume with a ValueError."""
    import csv
    from chronoclust_amd.app import append_to_file, drop_rows_after, write_file_header
    fn = str(tmp_path / "result.csv")
    write_file_header(fn, ["timepoint", "cumulative_size", "tracking_by_lineage"])
    append_to_file(fn, [[0, "10.0", "A"], [1, "12.5", "B"]])
    for torn in ("1,3", "", "x1,2.0,C", "1"):
        whole = open(fn, newline="").read()
        with open(fn, "a") as f:
            f.write(torn)
        drop_rows_after(fn, 1)
        with open(fn, newline="") as f:
            assert list(csv.reader(f)) == [["timepoint", "cumulative_size", "tracking_by_lineage"], ["0", "10.0", "A"],
                                           ["1", "12.5", "B"]]
        assert open(fn, newline="").read() == whole


def test_restore_of_a_foreign_or_incomplete_image_says_so(tmp_path):
    """restore_program_state on an .npz that save_program_state did not write (keys missing): a clear ValueError
    before anything touches the GPU, not a bare KeyError."""
    import pytest
    from chronoclust_amd import app
    d = str(tmp_path)
    np.savez(os.path.join(d, app.HDDSTREAM_OBJ + ".npz"), last_data_timestamp=np.int64(2))
    with pytest.raises(ValueError, match="holds no trackers"):
        app.restore_program_state(d, hddstream=None)
    for name in (app.TRACKER_HISTORICAL_ASSOC, app.TRACKER_LINEAGE):
        with open(os.path.join(d, name + ".pkl"), "wb") as f:
            f.write(b"x")
    with pytest.raises(ValueError, match="lacks dataset_size"):
        app.restore_program_state(d, hddstream=None)


def test_points_csv_formatter_writes_repr_bytes():
    """cc_format_points_csv (host side of the library, no GPU) against Python: every float as repr(float) - shortest
    round-trip digits, exponent notation below 1e-4 and from 1e16, '.0' on integral values -, ids, quoted labels, and the
    whole text equal to what the csv module / DataFrame.to_csv(index=False) write."""
    import csv
    import io
    import pandas as pd
    from chronoclust_amd import _lib
    rng = np.random.default_rng(1)
    specials = [0.0, -0.0, 1.0, -1.0, 0.1, 1e-4, 9.999e-5, 1e-5, 1.5e-5, 123456789.0, 1e15, 1e16, 9999999999999998.0,
                1.2345678901234567e16, 1e22, 1e23, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, 1 / 3, 100.0,
                12345.678, 0.000123, 4.35e-07, 29.90423, 123456789012345678.0, 2.5e-310, 1e100, 1e-100, 16.0]
    vals = np.array(specials + list(rng.uniform(-1, 1, 5000)) + list(rng.normal(0, 1, 5000) * 10.0 ** rng.integers(-30, 30, 5000))
                    + list(np.round(rng.uniform(0, 50, 5000), 5)) + list(rng.integers(-10 ** 6, 10 ** 6, 2000).astype(float)))
    text = _lib.format_points_csv(vals.reshape(-1, 1), 7, np.zeros(len(vals), np.int32), ["L"], threads=2, chunk=999).decode()
    assert text.split("\n")[:-1] == ["%d,L,%s" % (7 + i, repr(float(v))) for i, v in enumerate(vals)]
    n, d = 3000, 7
    Y = (rng.uniform(0, 1, (n, d)) - 0.1) / 0.013
    idx = rng.integers(-1, 3, n).astype(np.int32)
    names = ["A|1", "(B,C)", "C", "None"]
    quoted = []
    for s in names:
        b = io.StringIO()
        csv.writer(b, lineterminator="").writerow([s])
        quoted.append(b.getvalue())
    assert quoted[1] == '"(B,C)"'
    got = _lib.format_points_csv(Y, 0, idx, quoted, threads=3, chunk=512)
    cols = {"id": np.arange(n), "cluster_id": np.array(names, dtype=object)[idx]}
    cols.update({"m%d" % c: Y[:, c] for c in range(d)})
    f = io.StringIO()
    pd.DataFrame(cols).to_csv(f, index=False, header=False)
    assert got == f.getvalue().encode()
    assert _lib.format_points_csv(np.empty((0, 3)), 0, np.empty(0, np.int32), ["None"]) == b""
    # streamed to a file object: the same bytes, chunk by chunk in order, however many chunks are in flight
    import io
    for threads, chunk in ((1, 100), (3, 512), (16, 7)):
        sink = io.BytesIO()
        assert _lib.format_points_csv(Y, 0, idx, quoted, threads=threads, chunk=chunk, out=sink) == len(got)
        assert sink.getvalue() == got
    assert _lib.format_points_csv(np.empty((0, 3)), 0, np.empty(0, np.int32), ["None"], out=io.BytesIO()) == 0
    # NaN is an empty field, +-inf is spelled out: pandas' defaults
    Z = np.array([[1.5, np.nan, -np.inf], [np.nan, np.inf, 2.0]])
    f = io.StringIO()
    pd.DataFrame({"id": [0, 1], "cluster_id": ["C", "C"], "a": Z[:, 0], "b": Z[:, 1], "c": Z[:, 2]}).to_csv(f, index=False, header=False)
    assert _lib.format_points_csv(Z, 0, np.array([2, 2], np.int32), quoted) == f.getvalue().encode()
    # a label array that does not cover the rows is refused before the C side could read past it
    with pytest.raises(ValueError):
        _lib.format_points_csv(Y, 0, idx[:10], quoted)
    with pytest.raises(ValueError):
        _lib.format_points_csv(Y, 0, np.empty(0, np.int32), quoted)
    with pytest.raises(ValueError):
        _lib.format_points_csv(Y[0], 0, idx[:1], quoted)


def test_rounded_weights_equal_the_reference_expression():
    """chronoclust_amd.clustering.hddstream.rounded_weights against app.py:184 evaluated per value, on integers,
    decayed weights, exact and near half-way cases (x.y5 decimal strings whose doubles lie on either side)."""
    from decimal import ROUND_HALF_UP, Decimal
    import numpy as np
    from chronoclust_amd.clustering.hddstream import rounded_weights
    rng = np.random.default_rng(0)
    w = np.concatenate([
        rng.integers(0, 5000, 2000).astype(np.float64),
        rng.integers(0, 5000, 2000) * 2.0 ** (-rng.integers(1, 12, 2000) * 0.5),      # decayed weights
        rng.uniform(0, 3000, 4000),
        np.round(rng.uniform(0, 500, 4000), 2),                                        # two decimals: many x.y5
        np.array([0.05, 0.15, 0.25, 0.35, 1.45, 2.675, 1.005, 10.05, 200.25, 0.04999999999999999, 0.0, 1e15, 123456789.25]),
        np.nextafter(np.round(rng.uniform(0, 500, 500), 1) + 0.05, 0), np.nextafter(np.round(rng.uniform(0, 500, 500), 1) + 0.05, 1e9),
    ])
    got = rounded_weights(w)
    exp = [Decimal(str(float(x))).quantize(Decimal('1.1'), rounding=ROUND_HALF_UP) for x in w]
    assert got == exp
    assert [str(g) for g in got] == [str(e) for e in exp]
    assert [g.as_tuple() for g in got] == [e.as_tuple() for e in exp]


def test_label_agreement_is_symmetric_and_id_free():
    import numpy as np
    from chronoclust_amd import multi
    a = np.array([5, 5, 7, 7, 7, 9])
    assert multi.label_agreement(a, a + 100) == 1.0
    assert multi.label_agreement(a, np.array([1, 1, 2, 2, 3, 3])) == multi.label_agreement(np.array([1, 1, 2, 2, 3, 3]), a)
    assert multi.label_agreement(a, np.zeros(6, int)) == 0.5  # everything merged into one cluster: the largest of three
    assert multi.label_agreement(np.arange(6), a) == 0.5      # everything split into singletons


def test_point_cluster_index_maps_points_through_microclusters_to_clusters():
    import numpy as np
    from chronoclust_amd import multi
    # pcores (id, uid): (10, 0), (11, 1), (12, 2), (13, 7); clusters: [10, 12] and [11]; pcore 13 is in no cluster;
    # uid 5 is an outlier MC
    got = multi.point_cluster_index(np.array([0, 1, 2, 5, 1, 7]), np.array([10, 11, 12, 13]), np.array([0, 1, 2, 7]),
                                    np.array([10, 12, 11]), np.array([0, 2, 3]))
    assert got.tolist() == [0, 1, 0, -1, 1, -1]
    assert multi.point_cluster_index(np.array([3, 4]), np.array([], int), np.array([], int), np.array([], int),
                                     np.array([0])).tolist() == [-1, -1]


@pytest.mark.parametrize("stmt", ["import chronoclust.clustering.predecon", "from chronoclust.clustering import predecon",
                                  "from chronoclust.utilities import mc_functions", "import chronoclust.objects.predecon_mc",
                                  "import chronoclust.utilities.predeconmc_functions"])
def test_reference_modules_without_counterpart_say_why(stmt):
    """chronoclust/clustering/predecon.py:22 & co. are kernels here (seam B4): importing them fails with a
    message that names the replacement, not with a bare ModuleNotFoundError."""
    import chronoclust  # noqa: F401
    with pytest.raises(ImportError, match="no counterpart in the MI355X build") as e:
        exec(stmt, {})
    assert not isinstance(e.value, ModuleNotFoundError)
    with pytest.raises(ModuleNotFoundError):
        exec("import chronoclust.no_such_module", {})


def test_mutable_microcluster_object_replays_the_reference_unit_tests():
    """chronoclust.objects.microcluster.Microcluster (the reference's mutable object, objects/microcluster.py:18-257): the
    scenarios and known answers of the reference's own tests/objects_test/unittest_microcluster.py."""
    import numpy as np
    from chronoclust.objects.microcluster import Microcluster
    m = Microcluster(cf1=np.zeros(3), cf2=np.zeros(3), cluster_centroids=[0.1, 0.2, 0.03], preferred_dimension_vector=[1.0, 15.0, 15.0])
    assert round(m.get_projected_dist_to_point([1.0, 0.5, 0.7]), 2) == 0.85
    m = Microcluster(cf1=np.zeros(3), cf2=np.zeros(3), cluster_centroids=[-0.1, 0.2, -0.03], preferred_dimension_vector=[1.0, 15.0, 15.0])
    assert round(m.get_projected_dist_to_point([1.0, 0.5, 0.7]), 2) == 1.25
    points = [[0.17550518, 0.50150137, 0.0715026, 0.46715915, 0.11825116], [0.09084978, 0.33935363, 0.06932869, 0.78185322, 0.62759489],
              [0.22507306, 0.02771729, 0.46630673, 0.75367467, 0.2201496], [0.26507548, 0.44774516, 0.28568398, 0.80777178, 0.12095075],
              [0.43343372, 0.35738624, 0.4001447, 0.89195078, 0.29652304], [0.48627326, 0.52784397, 0.22927219, 0.801923, 0.07897944],
              [0.31972963, 0.29667314, 0.20070554, 0.31300255, 0.4958211], [0.05191981, 0.76440696, 0.0478006, 0.0201296, 0.25368318],
              [0.18290483, 0.65387882, 0.174167, 0.21822311, 0.2230557], [0.87574659, 0.77501901, 0.21127804, 0.15939672, 0.6381301]]
    k = 15
    for delta_sq, expected in ((0.01, [1, 1, 1, 1, 1]), (0.05, [1, k, k, 1, k]), (0.1, [k, k, k, k, k])):
        mc = Microcluster(cf1=np.zeros(5), cf2=np.zeros(5))
        for idx, p in enumerate(points):
            mc.add_new_point(np.array(p), 0, idx)
            mc.update_preferred_dimensions(delta_sq, k)
        np.testing.assert_equal(mc.preferred_dimension_vector, np.array(expected))
        assert list(mc.points.keys()) == list(range(10)) and mc.cumulative_weight == 10
    cf1 = [0.68756544, 0.96853843, 0.41156436, 0.13236377, 0.12836222, 0.55662013, 0.9671396, 0.99469293, 0.86402299, 0.90838236,
           0.52934492, 0.37423623, 0.02787237, 0.35216188, 0.96222637, 0.09291304, 0.08972414, 0.76429683, 0.78941125, 0.53722776]
    cf2 = [4.72746229e-01, 9.38066699e-01, 1.69385220e-01, 1.75201686e-02, 1.64768583e-02, 3.09825969e-01, 9.35359004e-01,
           9.89414034e-01, 7.46535721e-01, 8.25158518e-01, 2.80206042e-01, 1.40052759e-01, 7.76869185e-04, 1.24017991e-01,
           9.25879595e-01, 8.63283346e-03, 8.05042136e-03, 5.84149644e-01, 6.23170114e-01, 2.88613669e-01]
    pref = [1, 1, 16, 16, 1, 16, 16, 16, 16, 16, 1, 16, 1, 16, 16, 16, 1, 1, 1, 16]
    mc = Microcluster(cf1=np.array(cf1), cf2=np.array(cf2), preferred_dimension_vector=np.array(pref), cumulative_weight=20)
    assert abs(mc.calculate_projected_radius_squared() - 0.1551429607662637) < 1e-10
    for args, expected in (((0.1, 1, 20), False), ((0.2, 30, 20), False), ((0.2, 1, 2), False), ((0.1, 30, 20), False),
                           ((0.2, 30, 2), False), ((0.1, 1, 2), False), ((0.2, 1, 20), True), ((0.1552, 20, 12), True)):
        assert mc.is_core(*args) is expected, args
    one = Microcluster(cf1=np.zeros(len(cf1)), cf2=np.zeros(len(cf1)))
    one.add_new_point(np.array(cf1), 0, 0)
    clone = one.get_copy()
    np.testing.assert_almost_equal(clone.CF1, cf1)
    np.testing.assert_almost_equal(clone.CF2, cf2)
    assert clone.cumulative_weight == 1 and clone.points == {}
    more = one.get_copy_with_new_point(np.array(cf1), 0.05, 4)
    assert more.cumulative_weight == 2 and one.cumulative_weight == 1 and list(more.preferred_dimension_vector) == [4] * len(cf1)

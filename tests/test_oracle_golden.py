"""Pins the CPU oracle (oracle/chrono_oracle.c) against vectors produced by the upstream reference.

Known answers are the reference's own unit-test constants
(chronoclust/tests/objects_test/unittest_microcluster.py, clustering_test/unittest_predecon.py,
clustering_test/unittest_hddstream.py); state dumps come from tests/golden/make_golden.py.
"""
import os

import numpy as np
import pytest

import scenarios
from golden_util import GOLDEN, StateDump, assert_tables_equal, blob_inputs
from oracle import oracle as O

CF1_20 = [0.68756544, 0.96853843, 0.41156436, 0.13236377, 0.12836222, 0.55662013, 0.9671396, 0.99469293, 0.86402299,
          0.90838236, 0.52934492, 0.37423623, 0.02787237, 0.35216188, 0.96222637, 0.09291304, 0.08972414, 0.76429683,
          0.78941125, 0.53722776]
CF2_20 = [4.72746229e-01, 9.38066699e-01, 1.69385220e-01, 1.75201686e-02, 1.64768583e-02, 3.09825969e-01,
          9.35359004e-01, 9.89414034e-01, 7.46535721e-01, 8.25158518e-01, 2.80206042e-01, 1.40052759e-01,
          7.76869185e-04, 1.24017991e-01, 9.25879595e-01, 8.63283346e-03, 8.05042136e-03, 5.84149644e-01,
          6.23170114e-01, 2.88613669e-01]
PREF_20 = [1, 1, 16, 16, 1, 16, 16, 16, 16, 16, 1, 16, 1, 16, 16, 16, 1, 1, 1, 16]


def test_projected_distance_known_answers():  # unittest_microcluster.py:10-32
    assert round(O.projected_distance([0.1, 0.2, 0.03], [1.0, 15.0, 15.0], [1.0, 0.5, 0.7]), 2) == 0.85
    assert round(O.projected_distance([-0.1, 0.2, -0.03], [1.0, 15.0, 15.0], [1.0, 0.5, 0.7]), 2) == 1.25


def test_radius_squared_known_answer():  # unittest_microcluster.py:82-104
    assert abs(O.projected_radius_sq(CF1_20, CF2_20, PREF_20, 20) - 0.1551429607662637) < 1e-10


def test_is_core_truth_table():  # unittest_microcluster.py:124-164
    args = (CF1_20, CF2_20, PREF_20, 20)
    assert not O.is_core(*args, 0.1, 1, 20)
    assert not O.is_core(*args, 0.2, 30, 20)
    assert not O.is_core(*args, 0.2, 1, 2)
    assert not O.is_core(*args, 0.1, 30, 20)
    assert not O.is_core(*args, 0.2, 30, 2)
    assert not O.is_core(*args, 0.1, 1, 2)
    assert O.is_core(*args, 0.2, 1, 20)
    assert O.is_core(*args, 0.1552, 20, 12)


def test_update_preferred_dimensions():  # unittest_microcluster.py:34-80
    pts = np.array([[0.17550518, 0.50150137, 0.0715026, 0.46715915, 0.11825116],
                    [0.09084978, 0.33935363, 0.06932869, 0.78185322, 0.62759489],
                    [0.22507306, 0.02771729, 0.46630673, 0.75367467, 0.2201496],
                    [0.26507548, 0.44774516, 0.28568398, 0.80777178, 0.12095075],
                    [0.43343372, 0.35738624, 0.4001447, 0.89195078, 0.29652304],
                    [0.48627326, 0.52784397, 0.22927219, 0.801923, 0.07897944],
                    [0.31972963, 0.29667314, 0.20070554, 0.31300255, 0.4958211],
                    [0.05191981, 0.76440696, 0.0478006, 0.0201296, 0.25368318],
                    [0.18290483, 0.65387882, 0.174167, 0.21822311, 0.2230557],
                    [0.87574659, 0.77501901, 0.21127804, 0.15939672, 0.6381301]])
    cf1, cf2 = np.zeros(5), np.zeros(5)
    for p in pts:
        cf1, cf2 = cf1 + p, cf2 + p * p
    k = 15
    assert O.update_pref(cf1, cf2, 10, 0.01, k).tolist() == [1, 1, 1, 1, 1]
    assert O.update_pref(cf1, cf2, 10, 0.05, k).tolist() == [1, k, k, 1, k]
    assert O.update_pref(cf1, cf2, 10, 0.1, k).tolist() == [k, k, k, k, k]


def test_predecon_function_known_answers():  # unittest_predecon.py:8-47
    assert abs(O.euclidean([1, 5, 6, 3, 2], [6, 4, 6, 4, 2]) - 5.196152422706632) < 1e-7
    assert abs(O.weighted_dist_sq([15, 1, 1, 15], [0.1, 4.5, 4.2, 3.0], [1.1, 4.3, 2.2, 4.1]) - 37.19) < 1e-7
    point = [0.187, 0.922, 0.896, 0.098, 0.707, 0.626, 0.447, 0.588, 0.752, 0.041]
    neighbours = np.array([
        [0.873, 0.179, 0.585, 0.036, 0.051, 0.708, 0.485, 0.75, 0.665, 0.019],
        [0.218, 0.791, 0.451, 0.061, 0.197, 0.083, 0.453, 0.538, 0.136, 0.046],
        [0.314, 0.119, 0.153, 0.336, 0.174, 0.125, 0.02, 0.752, 0.89, 0.147],
        [0.21, 0.681, 0.018, 0.503, 0.081, 0.612, 0.395, 0.458, 0.071, 0.992],
        [0.26, 0.59, 0.788, 0.063, 0.466, 0.702, 0.387, 0.204, 0.91, 0.888],
        [0.775, 0.173, 0.92, 0.854, 0.034, 0.511, 0.933, 0.237, 0.375, 0.891],
        [0.441, 0.021, 0.142, 0.754, 0.121, 0.626, 0.661, 0.618, 0.967, 0.345],
        [0.457, 0.708, 0.322, 0.715, 0.075, 0.212, 0.481, 0.347, 0.935, 0.234],
        [0.516, 0.052, 0.745, 0.137, 0.764, 0.515, 0.888, 0.948, 0.362, 0.912],
        [0.287, 0.385, 0.658, 0.735, 0.354, 0.317, 0.321, 0.995, 0.071, 0.864]])
    expected = [0.109, 0.385, 0.261, 0.202, 0.275, 0.085, 0.068, 0.07, 0.173, 0.392]
    got = [O.variance_along_dimension(point[i], neighbours[:, i]) for i in range(10)]
    np.testing.assert_almost_equal(np.round(got, 3), expected)


def test_hddstream_dataset_dependent_parameters():  # unittest_hddstream.py:10-41
    cfg = {"beta": 0.5, "delta": 0.3, "epsilon": 10, "lambda": 1, "k": 40, "mu": 0.1, "pi": 0, "omicron": 0.001,
           "upsilon": 3}
    h = O.OracleHDDStream(cfg)
    h.online_microcluster_maintenance(np.random.rand(10, 2), 0)
    assert (h.pi, h.mu, h.omicron, h.upsilon) == (2, 1, 0, 30)
    h.online_microcluster_maintenance(np.random.rand(30, 2), 1)
    assert (h.pi, h.mu, h.omicron, h.upsilon) == (2, 3, 0.01, 30)


def test_hddstream_upgrade_when_dimension_loses_preference():  # unittest_hddstream.py:43-88
    cfg = {"beta": 0.5, "delta": 0.3, "epsilon": 10, "lambda": 1, "k": 40, "mu": 1, "pi": 2, "omicron": 1,
           "upsilon": 1}
    h = O.OracleHDDStream(cfg)
    h.online_microcluster_maintenance(np.array([[0.966970507, 0.185628831, 0.861853663],
                                                [0.557335192, 0.324320201, 0.495929691],
                                                [0.698145385, 0.222617485, 0.83843284],
                                                [0.466592479, 0.557335192, 0.993609292]]), 0)
    assert len(h.table(O.PCORE)["id"]) == 0
    out = h.table(O.OUTLIER)
    assert len(out["id"]) == 1 and out["pref"][0].tolist() == [40, 40, 40]
    h.online_microcluster_maintenance(np.array([[0.91259336, 0.16408931, 0.06039347]]), 0)
    pc = h.table(O.PCORE)
    assert len(pc["id"]) == 1 and len(h.table(O.OUTLIER)["id"]) == 0
    assert pc["pref"][0].tolist() == [40, 40, 1]


def _replay(dump, Xs, config, exact=True):
    h = O.OracleHDDStream(config)
    for t in range(dump.n_timepoints):
        X = Xs[t]
        h.online_microcluster_maintenance(X, int(dump.get(t, "daystamp")))
        par = dump.get(t, "params")
        assert (h.pi, h.mu, h.omicron) == (par[0], par[1], par[2])
        assert h.counters == (int(par[3]), int(par[4]))
        np.testing.assert_array_equal(h.labels_uid, dump.get(t, "labels_uid"), err_msg="labels t=%d" % t)
        assert_tables_equal(h.table(O.PCORE), dump, t, "pcore", exact)
        assert_tables_equal(h.table(O.OUTLIER), dump, t, "outlier", exact)
        got, exp = h.clusters, dump.clusters(t)
        assert len(got) == len(exp)
        for g, e in zip(got, exp):
            np.testing.assert_array_equal(g["members"], e["members"])  # merge order
            for key in ("w", "cf1", "cf2", "cen", "pref"):
                assert np.array_equal(np.asarray(g[key]), np.asarray(e[key])), (t, key)
    return h


def test_c1_state_matches_reference():
    dump = StateDump(os.path.join(GOLDEN, "c1", "hdd_state.npz"))
    Xs = [dump.get(t, "X") for t in range(dump.n_timepoints)]
    _replay(dump, Xs, scenarios.params_to_config(scenarios.C1_PARAMS))


def _c1_sample_run_inputs(dump):
    """The scaled inputs are those of c1/ (same files, same scaler); the dump only keeps their hashes."""
    import hashlib
    c1 = StateDump(os.path.join(GOLDEN, "c1", "hdd_state.npz"))
    Xs = [c1.get(t, "X") for t in range(c1.n_timepoints)]
    for t, X in enumerate(Xs):
        sha = np.frombuffer(hashlib.sha256(np.ascontiguousarray(X).tobytes()).digest(), dtype=np.uint8)
        assert (sha == dump.get(t, "xsha")).all()
    return Xs


def test_c1_sample_run_state_matches_reference():
    """BASELINE.json config 1 literally (sample_run_script/sample_run.py:6-22, omicron 4.35e-6: other microcluster
    counts than c1/ from the fourth timepoint on - 10/99 and 15/124 instead of 10/101 and 15/128)."""
    dump = StateDump(os.path.join(GOLDEN, "c1_sample_run", "hdd_state.npz"))
    h = _replay(dump, _c1_sample_run_inputs(dump), scenarios.params_to_config(scenarios.SAMPLE_RUN_PARAMS))
    assert (len(h.table(O.PCORE)["id"]), len(h.table(O.OUTLIER)["id"])) == (15, 124)


def test_nocluster_state_matches_reference():
    dump = StateDump(os.path.join(GOLDEN, "nocluster", "hdd_state.npz"))
    Xs = [dump.get(t, "X") for t in range(dump.n_timepoints)]
    _replay(dump, Xs, scenarios.params_to_config(scenarios.NOCLUSTER_PARAMS))


@pytest.mark.parametrize("name", sorted(scenarios.BLOB_SCENARIOS))
def test_blob_state_matches_reference(name):
    path = os.path.join(GOLDEN, "blob_%s.npz" % name)
    dump = StateDump(path)
    Xs = blob_inputs(name, dump)
    _replay(dump, Xs, scenarios.params_to_config(scenarios.BLOB_SCENARIOS[name]["params"]))

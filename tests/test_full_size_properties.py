"""GPU tests at BASELINE.json's full single-GPU size (C2: 1 M x 20, 5 000 microclusters), where the oracle
would need minutes: size-independent properties of the exact algorithm, plus an oracle check of a prefix
(the sequential algorithm's decisions on the first m points do not depend on later points)."""
import numpy as np
import pytest

import pipeline_util as P
import scenarios

pytestmark = pytest.mark.gpu

N, D, G = 1_000_000, 20, 5000


@pytest.fixture(scope="module")
def c2():
    from chronoclust_amd.clustering.hddstream import HDDStream
    X = scenarios.make_blobs(42, N, D, G)
    cfg = scenarios.params_to_config(scenarios.blob_params(N))
    h = HDDStream(cfg)
    h.online_microcluster_maintenance(X, 0)
    return X, cfg, h


def test_c2_weight_conservation_and_label_counts(c2):
    X, cfg, h = c2
    pc, ol = h.table(0), h.table(1)
    w = np.concatenate([pc["w"], ol["w"]])
    uid = np.concatenate([pc["uid"], ol["uid"]])
    assert w.sum() == N  # every point adds weight 1 to exactly one microcluster (no decay at t = 0)
    u, counts = np.unique(h.labels_uid, return_counts=True)
    order = np.argsort(uid)
    assert np.array_equal(uid[order], u)
    assert np.array_equal(w[order], counts.astype(np.float64))
    assert len(pc["id"]) == G and len(ol["id"]) == 0
    assert sorted(pc["id"].tolist()) == list(range(G))


def test_c2_cf_vectors_are_ordered_sums_of_member_points(c2):
    """CF1 / CF2 of a microcluster = its points added one by one in arrival order (float64, no reassociation)."""
    X, cfg, h = c2
    pc = h.table(0)
    rng = np.random.default_rng(1)
    for pos in rng.choice(len(pc["id"]), 25, replace=False):
        rows = np.nonzero(h.labels_uid == pc["uid"][pos])[0]
        cf1, cf2 = np.zeros(D), np.zeros(D)
        for r in rows:
            cf1 = cf1 + X[r]
            cf2 = cf2 + X[r] * X[r]
        assert np.array_equal(cf1, pc["cf1"][pos]) and np.array_equal(cf2, pc["cf2"][pos])
        assert np.array_equal(cf1 / len(rows), pc["cen"][pos])


def test_c2_labels_do_not_depend_on_window_or_segments(c2):
    from chronoclust_amd.clustering.hddstream import HDDStream
    X, cfg, h = c2
    for tuning in (dict(window=1536, segments=64, rounds=2, lookahead=3), dict(window=8192, segments=256, rounds=4, lookahead=2),
                   dict(window=16384, segments=128, lookahead=3), dict(window=32768, lookahead=3), dict(window=12288, lookahead=2)):
        g = HDDStream(cfg, tuning=tuning)
        g.online_microcluster_maintenance(X, 0)
        assert np.array_equal(g.labels_uid, h.labels_uid)
        for kind in (0, 1):
            a, b = g.table(kind), h.table(kind)
            for key in ("id", "uid", "w", "cf1", "cf2", "cen", "pref"):
                assert np.array_equal(a[key], b[key])
        assert [c.members_in_merge_order for c in g.final_clusters] == [c.members_in_merge_order for c in h.final_clusters]


@pytest.mark.parametrize("env", [dict(CHRONOCLUST_HIP_PRUNE=0), dict(CHRONOCLUST_HIP_PRUNE=2, CHRONOCLUST_HIP_SCANA=2),
                                 dict(CHRONOCLUST_HIP_PRUNE=2, CHRONOCLUST_HIP_GUESS=0), dict(CHRONOCLUST_HIP_LINK=0, CHRONOCLUST_HIP_NODIRTY=0)],
                         ids=["plain_scans", "pruned_forced_split", "seeded_thresholds_only", "no_links_dirty_scans_always"])
def test_c2_pruned_steady_state_equals_plain_scans_at_full_size(c2, env):
    """The pruned chain at C2's own size - 5 000 rows, full windows, guessed / lean thresholds: the 800 000
    points behind the start-up phase - against the PLAIN scan of every window (CHRONOCLUST_HIP_PRUNE=0), against the pruned
    chain forced from the first window on in its two-kernel form, and against seeded thresholds only: bit for bit.  (The
    tunings of the test above all prune; a threshold bug that only shows at full size would be invariant under them.
    tools/full_oracle.py runs the whole stream through the oracle - minutes of CPU, recorded under profiles/.)"""
    from chronoclust_amd.clustering.hddstream import HDDStream
    X, cfg, h = c2
    assert h.stats()["scan_p_launches"] >= 15  # (the default run's steady state is pruned: ~800 000 points in windows of 49 152)
    with P.knobs(**env):
        g = HDDStream(cfg)
    g.online_microcluster_maintenance(X, 0)
    st = g.stats()
    if env.get("CHRONOCLUST_HIP_PRUNE") == 0:
        assert st["scan_p_launches"] == 0 and st["scan_u_launches"] > 20
    if env.get("CHRONOCLUST_HIP_PRUNE") == 2:
        assert st["scan_p_launches"] == st["scan_u_launches"]
    if env.get("CHRONOCLUST_HIP_LINK") == 0:
        assert st["link_launches"] == 0 and h.stats()["link_launches"] > 0
    assert np.array_equal(g.labels_uid, h.labels_uid)
    for kind in (0, 1):
        a, b = g.table(kind), h.table(kind)
        for key in ("id", "uid", "w", "cf1", "cf2", "cen", "pref"):
            assert np.array_equal(a[key], b[key]), (kind, key)
    assert [c.members_in_merge_order for c in g.final_clusters] == [c.members_in_merge_order for c in h.final_clusters]


def test_c2_prefix_matches_oracle(c2):
    from oracle import oracle as O
    X, cfg, h = c2
    m = 120_000
    o = O.OracleHDDStream(cfg)
    o.set_dataset_dependent_parameters(X)  # thresholds of the full timepoint (mu = mu_cfg * N)
    o.online_microcluster_maintenance(X[:m], 0, reset_param=False, offline=False)
    assert np.array_equal(o.labels_uid, h.labels_uid[:m])


def test_three_timepoints_100k_with_churn_against_oracle():
    """C3-shaped (several timepoints with decay, drift, retired and new blobs), sized so the oracle takes seconds."""
    from chronoclust_amd.clustering.hddstream import HDDStream
    from oracle import oracle as O
    sc = dict(seed=5, n=100_000, d=20, g=1500, sigma=0.01, timepoints=3, drift=0.01, churn=0.08)
    params = scenarios.blob_params(sc["n"], param_omicron=0.00002, param_lambda=2)
    cfg = scenarios.params_to_config(params)
    Xs = scenarios.make_blob_timepoints(sc, raw=True)
    h, o = HDDStream(cfg), O.OracleHDDStream(cfg)
    for t, X in enumerate(Xs):
        h.online_microcluster_maintenance(X, t)
        o.online_microcluster_maintenance(X, t)
        assert np.array_equal(h.labels_uid, o.labels_uid)
        assert (h.pcore_MC_last_id, h.outlier_MC_last_id) == o.counters
        for kind in (0, 1):
            a, b = h.table(kind), o.table(kind)
            for key in ("id", "uid", "w", "cf1", "cf2", "cen", "pref"):
                assert np.array_equal(a[key], b[key]), (t, kind, key)
        got, exp = h.final_clusters, o.clusters
        assert [c.members_in_merge_order for c in got] == [[int(x) for x in c["members"]] for c in exp]
        for g_, e_ in zip(got, exp):
            assert g_.cumulative_weight == e_["w"] and np.array_equal(g_.cluster_centroids, e_["cen"])
    assert len(h.table(1)["id"]) > 0 or h.outlier_MC_last_id > len(h.table(0)["id"])  # downgrade / delete paths ran

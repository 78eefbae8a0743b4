"""GPU tests of the other BASELINE.json shapes at full single-GPU size, where the oracle would need hours:

  C3         5 timepoints x 1 M x 20 with drift, churn, decay, downgrade / delete and both trackers
  C4-shaped  5 M x 14, 2 000 microclusters (one timepoint of the WNV-shaped config; its 8-GPU half is the driver's)
  C5-shaped  2 M x 40, 50 000 microclusters (the per-GPU share of the stress config's points with its full table)

Each is checked through size-independent properties of the exact algorithm (weight conservation point by point,
CF vectors = ordered sums of member points, results independent of window size / segments / lookahead - labels,
tables AND the lineage / association strings), against the oracle on a prefix of the first timepoint (the
sequential algorithm's decisions on the first m points do not depend on later ones), and - C3 - the association
argmin of a whole timepoint (5 000 x 5 000 pairs) against the oracle's."""
import numpy as np
import pytest

import pipeline_util as P
import scenarios

pytestmark = pytest.mark.gpu


def _oracle_prefix(cfg, X, m, labels_uid):
    from oracle import oracle as O
    o = O.OracleHDDStream(cfg)
    o.set_dataset_dependent_parameters(X)  # thresholds of the full timepoint (mu = mu_cfg * N)
    o.online_microcluster_maintenance(X[:m], 0, reset_param=False, offline=False)
    assert np.array_equal(o.labels_uid, labels_uid[:m])
    return o


# ---------------------------------------------------------------------------------------------------------
# C3
# ---------------------------------------------------------------------------------------------------------

C3 = dict(seed=42, n=1_000_000, d=20, g=5000, sigma=0.01, timepoints=5, drift=0.01, churn=0.02)
# lambda = 2: a retired blob's microcluster (weight ~200) decays 200 -> 50 -> 12.5 -> 3.1: below beta * mu = 10 at
# its third timepoint (downgrade) and below omicron * N = 20 (delete, with the skip-next quirk of hddstream.py:545-549)
C3_PARAMS = scenarios.blob_params(C3["n"], param_lambda=2, param_omicron=0.00002)


@pytest.fixture(scope="module")
def c3():
    Xs = scenarios.make_blob_timepoints(C3, raw=True)
    cfg = scenarios.params_to_config(C3_PARAMS)
    return Xs, cfg, P.run_pipeline(Xs, cfg)


def test_c3_weights_counts_and_ordered_sums(c3):
    Xs, cfg, res = c3
    rng = np.random.default_rng(3)
    f = 2 ** (-cfg["lambda"] * 1)
    checked = 0
    for t, rec in enumerate(res):
        P.check_weights(rec, C3["n"], f)
        checked += P.check_cf_ordered_sums(Xs[t], rec, rng, samples=6)
        assert len(rec["pcore"]["id"]) >= C3["g"] * 0.9
        assert len(rec["rows"]) >= C3["g"] * 0.9  # well-separated blobs: one cluster per live blob
    assert checked >= 12
    # the churn really exercised the timestep boundary: outlier ids beyond the blobs were handed out, microclusters
    # were downgraded / deleted, and the trackers produced both inherited and fresh lineage letters
    assert res[-1]["counters"][1] > C3["g"]
    uids0 = set(res[0]["pcore"]["uid"].tolist())
    assert len(uids0 - set(res[-1]["pcore"]["uid"].tolist()) - set(res[-1]["outlier"]["uid"].tolist())) > 0
    ids0 = {r[3] for r in res[0]["rows"]}
    ids4 = {r[3] for r in res[-1]["rows"]}
    assert len(ids0 & ids4) > C3["g"] // 2 and len(ids4 - ids0) > 0
    assert all(r[4] == "None" for r in res[0]["rows"]) and all(r[4] != "None" for r in res[1]["rows"])


@pytest.mark.parametrize("tuning", [dict(window=8192, segments=128, rounds=4, lookahead=2),
                                    dict(window=32768, lookahead=3)])
def test_c3_results_do_not_depend_on_window_or_lookahead(c3, tuning):
    Xs, cfg, res = c3
    P.same_results(P.run_pipeline(Xs, cfg, tuning=tuning), res)


def test_c3_pruned_scans_equal_plain_scans(c3):
    """All five timepoints with CHRONOCLUST_HIP_PRUNE=0 (every window's snapshot scan the plain k_scan_u): labels, tables,
    merge order, lineage and association strings equal the default run's, whose steady state is pruned."""
    Xs, cfg, res = c3
    assert all(r["stats"]["scan_p_launches"] > 0 for r in res)
    with P.knobs(CHRONOCLUST_HIP_PRUNE=0):
        plain = P.run_pipeline(Xs, cfg)
    assert all(r["stats"]["scan_p_launches"] == 0 for r in plain)
    P.same_results(plain, res)


def test_c3_array_built_cluster_records_equal_object_built(c3):
    """HDDStream.cluster_records (arrays) against the reference's per-object construction (app.py:179-190) at
    5 000 clusters: same weights, pcore id order, lineage and association strings."""
    Xs, cfg, res = c3
    P.same_results(P.run_pipeline(Xs[:3], cfg, object_records=True), res[:3])


def test_c3_first_timepoint_prefix_matches_oracle(c3):
    Xs, cfg, res = c3
    _oracle_prefix(cfg, Xs[0], 100_000, res[0]["labels_uid"])


def test_c3_association_argmin_of_a_timepoint_matches_oracle(c3):
    """cluster_tracker.py:127-141 at full size: every current pcore against every previous pcore."""
    from oracle import oracle as O
    Xs, cfg, res = c3
    calls = res[2]["assoc_calls"]
    assert len(calls) == 1
    c = calls[0]
    assert c["cur_cen"].shape[0] >= C3["g"] * 0.9 and c["prev_cen"].shape[0] >= C3["g"] * 0.9
    idx, dist = O.assoc_argmin(c["cur_cen"], c["cur_pref"], c["prev_cen"])
    assert np.array_equal(idx, c["idx"])
    assert np.array_equal(dist, c["dist"])


# ---------------------------------------------------------------------------------------------------------
# C4-shaped and C5-shaped
# ---------------------------------------------------------------------------------------------------------

SHAPES = {
    # name: (points, dim, blobs, oracle prefix, tunings to compare with the default)
    "C4": (5_000_000, 14, 2000, 60_000, [dict(window=8192, segments=128, lookahead=2), dict(window=32768, lookahead=3)]),
    "C5": (2_000_000, 40, 50_000, 40_000, [dict(window=12288, segments=256, lookahead=2)]),
}


@pytest.fixture(scope="module", params=sorted(SHAPES))
def shape(request):
    n, d, g, m, tunings = SHAPES[request.param]
    X = scenarios.make_blobs(42, n, d, g)
    cfg = scenarios.params_to_config(scenarios.blob_params(n))
    res = P.run_pipeline([X], cfg)
    return request.param, X, cfg, res


def test_shape_weights_counts_and_ordered_sums(shape):
    name, X, cfg, res = shape
    n, d, g, m, _ = SHAPES[name]
    rec = res[0]
    P.check_weights(rec, n)
    assert P.check_cf_ordered_sums(X, rec, np.random.default_rng(5), samples=10) == 10
    pc, ol = rec["pcore"], rec["outlier"]
    if n >= 10 * g * 10:
        # every blob received far more than the 10 points a promotion takes: G pcores, nothing left behind
        assert len(pc["id"]) == g and len(ol["id"]) == 0
        assert sorted(pc["id"].tolist()) == list(range(g))
    else:
        # C5-shaped: 40 points per blob on average - a few blobs stay below the promotion threshold
        assert len(pc["id"]) + len(ol["id"]) == g and len(pc["id"]) > 0.99 * g
    # blobs are far apart: one cluster per core pcore (a pcore below mu = 2 * beta * mu is not core: no cluster)
    n_core = int((pc["w"] >= cfg["mu"] * n).sum())
    assert len(rec["rows"]) == n_core
    assert [len(mm) for mm in rec["members"]] == [1] * n_core


def test_shape_results_do_not_depend_on_window_or_lookahead(shape):
    name, X, cfg, res = shape
    for tuning in SHAPES[name][4]:
        P.same_results(P.run_pipeline([X], cfg, tuning=tuning), res)


def test_shape_pruned_scans_equal_plain_scans(shape):
    """The whole stream of either shape - 5 M x 14 / 2 000 rows, 2 M x 40 / 50 000 rows - with CHRONOCLUST_HIP_PRUNE=0 (every
    window's snapshot scan the plain k_scan_u) against the default run, whose settled stretch is pruned: bit for bit."""
    name, X, cfg, res = shape
    with P.knobs(CHRONOCLUST_HIP_PRUNE=0):
        plain = P.run_pipeline([X], cfg)
    assert plain[0]["stats"]["scan_p_launches"] == 0 and res[0]["stats"]["scan_p_launches"] > 0
    P.same_results(plain, res)


def test_c5_shape_pruned_equals_plain_on_the_settled_table():
    """50 000 rows x 40 dimensions, full windows, lookahead: the split pruned scan (k_scan_a + k_scan_p<MASKED>, the default from
    10 000 rows on), the one-kernel form and the PLAIN scan over the same 262 144 fresh points on the same settled table -
    bit for bit.  The table is settled by the default path (2 M points); the other handles take it over through the
    checkpoint arrays (cc_inject_bulk: another row numbering, the same lists), so the plain scans - 20 us per point at this
    shape - only run over the stretch that is compared.  (The whole 2 M stream with plain scans: tools/full_oracle.py c5plain.)"""
    from chronoclust_amd.clustering.hddstream import HDDStream
    n, d, g, m = 2_000_000, 40, 50_000, 262_144
    X = scenarios.make_blobs(42, n, d, g)
    cfg = scenarios.params_to_config(scenarios.blob_params(n))
    centres = np.random.default_rng(42).uniform(0.1, 0.9, (g, d))  # (make_blobs(42, ..) draws its centres first)
    rng = np.random.default_rng(4242)
    Y = np.ascontiguousarray(np.clip(centres[rng.integers(0, g, m)] + rng.normal(0.0, 0.01, (m, d)), 0.0, 1.0))
    base = HDDStream(cfg)
    base.online_microcluster_maintenance(X, 0)
    state = base.get_state()
    base.online_microcluster_maintenance(Y, 0, reset_param=False)
    st0 = base.stats()
    runs = {}
    for name, env in (("plain", dict(CHRONOCLUST_HIP_PRUNE=0)), ("one kernel", dict(CHRONOCLUST_HIP_PRUNE=2, CHRONOCLUST_HIP_SCANA=0)),
                      ("split", dict(CHRONOCLUST_HIP_PRUNE=2, CHRONOCLUST_HIP_SCANA=2))):
        with P.knobs(**env):
            h = HDDStream(cfg)
        h.set_state(state)
        h._set_dataset_dependent_parameters(X)  # (mu of the 2 M timepoint)
        s0 = h.stats()
        h.online_microcluster_maintenance(Y, 0, reset_param=False)
        s1 = h.stats()
        runs[name] = (s1["scan_p_launches"] - s0["scan_p_launches"], s1["scan_u_launches"] - s0["scan_u_launches"])
        assert np.array_equal(h.labels_uid, base.labels_uid), name
        for kind in (0, 1):
            a, b = h.table(kind), base.table(kind)
            for key in ("id", "uid", "w", "cf1", "cf2", "cen", "pref"):
                assert np.array_equal(a[key], b[key]), (name, kind, key)
        assert [c.members_in_merge_order for c in h.final_clusters] == [c.members_in_merge_order for c in base.final_clusters]
    assert runs["plain"][0] == 0 and runs["plain"][1] >= m // 49152  # (49 152: the largest window, the default since round 6)
    assert runs["one kernel"][0] == runs["one kernel"][1] > 0 and runs["split"][0] == runs["split"][1] > 0
    assert st0["scan_p_launches"] > 0 and st0["rows"] == g


def test_shape_prefix_matches_oracle(shape):
    """The first m points against the oracle: while the table fills up (0 -> ~35 000 rows at C5's shape) this is
    the creation / promotion regime of the d = 14 / d = 40 kernels."""
    name, X, cfg, res = shape
    _oracle_prefix(cfg, X, SHAPES[name][3], res[0]["labels_uid"])

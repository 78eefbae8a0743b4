"""BASELINE.json configs 4 and 5 at their CONFIGURED size, on one GPU (which holds both), plus their 8-rank forms as
in-process groups on that GPU:

  C5   1 timepoint, 50 M x 40, 50 000 microclusters (16 GB of points + their dimension-major copy in HBM)
  C4   8 timepoints x 5 M x 14, 2 000 microclusters, drift + churn + decay, both trackers (the app.run pipeline)
  C5-shaped, 8 ranks, exact      2 M x 40 / 50 000 microclusters, default split thresholds
  C4-shaped, 8 ranks, relaxed    5 M x 14 / 2 000 microclusters, events sharded, CF deltas all-reduced

The oracle would need days here; what is checked is what the exact algorithm guarantees at any size - every point adds
weight 1.0 to exactly one microcluster, CF vectors are the ordered sums of their points, labels / tables / lineage and
association strings do not depend on window size or lookahead -, the oracle on a prefix of the first timepoint (the
sequential algorithm's first m decisions do not depend on later points), and, for the groups, bit-equality with one GPU."""
import numpy as np
import pytest

import pipeline_util as P
import scenarios

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(1500)]


def _oracle_prefix(cfg, X, m, labels_uid):
    from oracle import oracle as O
    o = O.OracleHDDStream(cfg)
    o.set_dataset_dependent_parameters(X)  # thresholds of the full timepoint (mu = mu_cfg * N)
    o.online_microcluster_maintenance(np.ascontiguousarray(X[:m]), 0, reset_param=False, offline=False)
    assert np.array_equal(o.labels_uid, labels_uid[:m])


# ---------------------------------------------------------------------------------------------------------
# C5 at full size
# ---------------------------------------------------------------------------------------------------------

C5 = dict(n=50_000_000, d=40, g=50_000, seed=42)


def _host_memory_gib():
    import psutil
    avail = psutil.virtual_memory().available
    try:  # a container's own limit, where there is one
        with open("/sys/fs/cgroup/memory.max") as f:
            lim = f.read().strip()
        if lim != "max":
            with open("/sys/fs/cgroup/memory.current") as f:
                avail = min(avail, int(lim) - int(f.read().strip()))
    except OSError:
        pass
    return avail / 2 ** 30


@pytest.fixture(scope="module")
def c5():
    need = C5["n"] * C5["d"] * 8 / 2 ** 30 * 1.35
    have = _host_memory_gib()
    if have < need:
        pytest.skip("C5 at full size needs %.0f GiB of host memory for its input array, this box offers %.0f" % (need, have))
    X = scenarios.make_blobs_chunked(C5["seed"], C5["n"], C5["d"], C5["g"], threads=12)
    cfg = scenarios.params_to_config(scenarios.blob_params(C5["n"]))
    res = P.run_pipeline([X], cfg)
    return X, cfg, res


def test_c5_full_size_weights_counts_and_ordered_sums(c5):
    X, cfg, res = c5
    n, g = C5["n"], C5["g"]
    rec = res[0]
    P.check_weights(rec, n)
    assert P.check_cf_ordered_sums(X, rec, np.random.default_rng(5), samples=8) == 8
    pc, ol = rec["pcore"], rec["outlier"]
    # 1 000 points per blob: every blob far beyond the 10 points a promotion takes
    assert len(pc["id"]) == g and len(ol["id"]) == 0
    assert sorted(pc["id"].tolist()) == list(range(g))
    assert len(rec["rows"]) == g and [len(mm) for mm in rec["members"]] == [1] * g
    st = rec["stats"]
    assert st["points"] == n and st["rows"] == g
    print("C5 full size: online %.1f ms = %.2f M points/s, %d windows (%d scanned ahead), %d validation rounds" % (
        st["run_ms"], n / st["run_ms"] / 1e3, st["windows"], st["lookahead_windows"], st["rounds"]))


def test_c5_full_size_does_not_depend_on_window_or_lookahead(c5):
    X, cfg, res = c5
    P.same_results(P.run_pipeline([X], cfg, tuning=dict(window=12288, segments=256, lookahead=2)), res)


def test_c5_full_size_pruned_scans_equal_plain_scans(c5):
    """50 M x 40, 50 000 microclusters: every window's snapshot scan the plain k_scan_u (CHRONOCLUST_HIP_PRUNE=0) against the
    default run, 97 % of whose windows are pruned two-kernel chains on the full table - bit for bit."""
    X, cfg, res = c5
    with P.knobs(CHRONOCLUST_HIP_PRUNE=0):
        plain = P.run_pipeline([X], cfg)
    assert plain[0]["stats"]["scan_p_launches"] == 0 and res[0]["stats"]["scan_p_launches"] > 1000
    P.same_results(plain, res)
    print("C5 full size with plain scans: online %.1f ms" % plain[0]["stats"]["run_ms"])


def test_c5_full_size_prefix_matches_oracle(c5):
    X, cfg, res = c5
    _oracle_prefix(cfg, X, 20_000, res[0]["labels_uid"])  # (the 40 000-point prefix of this shape: test_full_size_shapes.py)


def test_c5_chunks_regenerate_alone(c5):
    """(the generator of this file: a chunk is a function of (seed, chunk index) only)"""
    X, _, _ = c5
    centres = scenarios.blob_centres(C5["seed"], C5["d"], C5["g"])
    c = 37
    a = c * scenarios.BLOB_CHUNK
    assert np.array_equal(X[a:a + 1000], scenarios.blob_chunk(C5["seed"], c, scenarios.BLOB_CHUNK, C5["d"], centres)[:1000])


# ---------------------------------------------------------------------------------------------------------
# C4 at full size: 8 timepoints through the pipeline of app.run, trackers on
# ---------------------------------------------------------------------------------------------------------

C4 = dict(seed=44, n=5_000_000, d=14, g=2000, sigma=0.01, timepoints=8, drift=0.01, churn=0.02)
# lambda = 2: a retired blob's microcluster (weight ~2 500) decays by 4 per timepoint: 625, 156, 39, 9.8 - below
# beta * mu = 10 at its fourth boundary (downgrade) and below omicron * N = 20 at the same one (delete, with the
# skip-next quirk of hddstream.py:545-549): blobs retired at t = 1 .. 3 are gone by t = 5 .. 7
C4_PARAMS = scenarios.blob_params(C4["n"], param_lambda=2, param_omicron=0.000004)


@pytest.fixture(scope="module")
def c4():
    Xs = scenarios.make_blob_timepoints(C4, raw=True)
    cfg = scenarios.params_to_config(C4_PARAMS)
    return Xs, cfg, P.run_pipeline(Xs, cfg)


def test_c4_full_size_weights_counts_and_ordered_sums(c4):
    Xs, cfg, res = c4
    rng = np.random.default_rng(3)
    f = 2 ** (-cfg["lambda"] * 1)
    checked = 0
    for t, rec in enumerate(res):
        P.check_weights(rec, C4["n"], f)
        checked += P.check_cf_ordered_sums(Xs[t], rec, rng, samples=4)
        assert len(rec["pcore"]["id"]) >= C4["g"] * 0.95
        assert len(rec["rows"]) >= C4["g"] * 0.95  # well-separated blobs: one cluster per live blob
    assert checked >= 12
    # the timestep boundary did its work: microclusters of retired blobs were downgraded and deleted, new blobs got
    # fresh outlier ids and then pcore ids; lineage letters were inherited and new ones handed out; every cluster of a
    # later timepoint has an associate
    assert res[-1]["counters"][1] > C4["g"] and res[-1]["counters"][0] > C4["g"]
    uids0 = set(res[0]["pcore"]["uid"].tolist())
    gone = uids0 - set(res[-1]["pcore"]["uid"].tolist()) - set(res[-1]["outlier"]["uid"].tolist())
    assert len(gone) >= 40  # (the 40 blobs retired at t = 1 are deleted by t = 5)
    ids0 = {r[3] for r in res[0]["rows"]}
    ids7 = {r[3] for r in res[-1]["rows"]}
    assert len(ids0 & ids7) > C4["g"] // 2 and len(ids7 - ids0) > 0
    assert all(r[4] == "None" for r in res[0]["rows"])
    for t in range(1, C4["timepoints"]):
        assert all(r[4] != "None" for r in res[t]["rows"])
    for t, rec in enumerate(res):
        st = rec["stats"]
        print("C4 t=%d: online %.1f ms = %.1f M points/s, %d pcore / %d outlier microclusters, %d clusters" % (
            t, st["run_ms"], C4["n"] / st["run_ms"] / 1e3, len(rec["pcore"]["id"]), len(rec["outlier"]["id"]), len(rec["rows"])))


@pytest.mark.parametrize("tuning", [dict(window=8192, segments=128, rounds=4, lookahead=2), dict(window=32768, lookahead=3)])
def test_c4_full_size_does_not_depend_on_window_or_lookahead(c4, tuning):
    """labels, tables, merge order AND the lineage / association strings of all eight timepoints"""
    Xs, cfg, res = c4
    P.same_results(P.run_pipeline(Xs, cfg, tuning=tuning), res)


def test_c4_full_size_first_timepoint_prefix_matches_oracle(c4):
    Xs, cfg, res = c4
    _oracle_prefix(cfg, Xs[0], 60_000, res[0]["labels_uid"])


def test_c4_full_size_association_argmin_matches_oracle(c4):
    from oracle import oracle as O
    Xs, cfg, res = c4
    c = res[5]["assoc_calls"][0]
    assert c["cur_cen"].shape[0] >= C4["g"] * 0.95 and c["prev_cen"].shape[0] >= C4["g"] * 0.95
    idx, dist = O.assoc_argmin(c["cur_cen"], c["cur_pref"], c["prev_cen"])
    assert np.array_equal(idx, c["idx"]) and np.array_equal(dist, c["dist"])


# ---------------------------------------------------------------------------------------------------------
# the 8-rank forms, as in-process groups on one GPU
# ---------------------------------------------------------------------------------------------------------

def test_eight_ranks_exact_at_the_c5_shape():
    """2 M x 40, 50 000 microclusters, EIGHT ranks with the default thresholds (scan split once rows x d >= 400 000,
    offline pair matrices once there are 8 192 pcores): every rank bit-equal to one GPU - with pruned scans among the
    split ones (round 4: a group no longer trades the pruned scan for the row split)."""
    from test_sharded_local import run_group
    n, d, g = 2_000_000, 40, 50_000
    X = scenarios.make_blobs(42, n, d, g)
    cfg = scenarios.params_to_config(scenarios.blob_params(n))
    single = P.run_pipeline([X], cfg)
    res = run_group(8, [X], cfg, min_row_dims=-1, offline_min_rows=-1)
    for r in res:
        P.same_results(r, single)
        st = r[0]["stats"]
        assert 0 < st["sharded_windows"] <= st["windows"]
        assert st["scan_p_launches"] > 0
    print("C5-shaped, 8 ranks: %d windows, %d split, %d pruned launches of %d, %.2f %% of the sampled rows completed" % (
        st["windows"], st["sharded_windows"], st["scan_p_launches"], st["scan_u_launches"],
        100.0 * st["pruned_scan_full_rows"] / max(1, st["pruned_scan_rows"])))
    assert len(single[0]["pcore"]["id"]) + len(single[0]["outlier"]["id"]) == g


def test_eight_ranks_relaxed_at_the_c4_shape():
    """5 M x 14, 2 000 microclusters, events sharded over EIGHT ranks in super-steps of up to 65 536 points per rank:
    ranks bit-identical, every point labelled, weights = label counts, and the exact path's clusters."""
    from chronoclust_amd import multi
    from chronoclust_amd.clustering.hddstream import HDDStream
    from test_relaxed_local import _same_on_all_ranks, run_relaxed_group
    n, d, g = 5_000_000, 14, 2000
    X = scenarios.make_blobs(777, n, d, g)
    cfg = scenarios.params_to_config(scenarios.blob_params(n))
    res = run_relaxed_group(8, [X], cfg, 65536)
    _same_on_all_ranks(res)
    r = res[0][0]
    exact = HDDStream(cfg)
    exact.online_microcluster_maintenance(X, 0)
    assert (r["labels"] >= 0).all() and not (r["paths"] & 8).any()
    uid = np.concatenate([r["pcore"]["uid"], r["outlier"]["uid"]])
    w = np.concatenate([r["pcore"]["w"], r["outlier"]["w"]])
    u, counts = np.unique(r["labels"], return_counts=True)
    order = np.argsort(uid)
    assert np.array_equal(uid[order], u) and np.array_equal(w[order], counts.astype(np.float64)) and w.sum() == n
    assert g <= len(r["pcore"]["id"]) + len(r["outlier"]["id"]) <= 1.02 * g
    by_cluster = multi.label_agreement(r["point_cluster"], exact.point_cluster_index())
    by_mc = multi.label_agreement(r["labels"], exact.labels_uid)
    print("C4-shaped, 8 ranks relaxed: agreement by cluster %.6f, by microcluster %.6f, microclusters %d (exact %d), "
          "set aside %d, super-steps %d" % (by_cluster, by_mc, len(uid), g, r["rstats"]["deferred_points"],
                                            r["rstats"]["super_steps"]))
    assert by_cluster >= 0.999 and by_mc >= 0.98

"""Randomised GPU-vs-oracle parity over the parameter surface (beta, delta, epsilon, lambda, k, mu, pi, omicron,
upsilon), dimensionalities 1..64, window sizes, and degenerate inputs: exact ties (points on an integer grid,
duplicates), k < 1 and k = 1, delta = 0, pi < d, tiny and empty-ish timepoints.  Everything is compared bit for
bit after every timepoint."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    d = int(rng.choice([1, 2, 3, 4, 5, 8, 9, 17, 20, 33, 64]))
    n = int(rng.choice([1, 2, 17, 300, 1500, 4000]))
    g = int(rng.choice([1, 2, 5, 12, 40]))
    sigma = float(rng.choice([0.0, 0.001, 0.02, 0.08, 0.3]))
    grid = bool(rng.random() < 0.3)  # snap to a coarse grid: many exact duplicates and distance ties
    cfg = {
        "beta": float(rng.choice([0.1, 0.5, 0.9, 1.0])),
        "delta": float(rng.choice([0.0, 0.01, 0.05, 0.3, 1.0])),
        "epsilon": float(rng.choice([0.001, 0.03, 0.1, 0.5, 3.0])),
        "lambda": float(rng.choice([0.0, 0.5, 2.0, 5.0])),
        "k": float(rng.choice([0.5, 1.0, 2.0, 3.0, 4.0, 16.0, 40.0])),
        "mu": float(rng.choice([0.0005, 0.002, 0.01, 0.1])),
        "pi": int(rng.choice([0, 1, max(1, d - 1), d, d + 3])),
        "omicron": float(rng.choice([0.0, 1e-5, 1e-3, 0.05])),
        "upsilon": float(rng.choice([0.5, 1.0, 3.0, 6.5, 20.0])),
    }
    window = int(rng.choice([1, 5, 64, 700, 4096]))
    if window == 1:
        n = min(n, 300)
    timepoints = []
    centres = rng.uniform(0.1, 0.9, (g, d))
    for t in range(3):
        nt = max(1, int(n * rng.choice([1.0, 0.5, 0.1]))) if t else n
        lab = rng.integers(0, g, nt)
        X = np.clip(centres[lab] + rng.normal(0.0, 1.0, (nt, d)) * sigma, 0.0, 1.0)
        if grid:
            X = np.round(X * 8) / 8
        timepoints.append(np.ascontiguousarray(X))
        centres = np.clip(centres + rng.normal(0, 0.02, centres.shape), 0, 1)
    return cfg, window, timepoints


# lookahead scans: forced from the first window on (3) / library default, i.e. while windows commit in full (0) / off (2)
@pytest.mark.parametrize("lookahead", [3, 0, 2])
@pytest.mark.parametrize("seed", range(192))
def test_fuzz_case(seed, lookahead):
    from chronoclust_amd.clustering.hddstream import HDDStream
    from oracle import oracle as O
    cfg, window, Xs = _case(seed)
    if lookahead == 0 and seed % 3:
        pytest.skip("default mode on every third case")
    h = HDDStream(cfg, tuning=dict(window=window, lookahead=lookahead))
    o = O.OracleHDDStream(cfg)
    for t, X in enumerate(Xs):
        h.online_microcluster_maintenance(X, t)
        o.online_microcluster_maintenance(X, t)
        ctx = "seed %d t %d cfg %s window %d lookahead %d n %d d %d" % (seed, t, cfg, window, lookahead, len(X), X.shape[1])
        np.testing.assert_array_equal(h.labels_uid, o.labels_uid, err_msg=ctx)
        np.testing.assert_array_equal(h.labels_path, o.paths, err_msg=ctx)
        assert (h.pcore_MC_last_id, h.outlier_MC_last_id) == o.counters, ctx
        for kind in (0, 1):
            a, b = h.table(kind), o.table(kind)
            for key in ("id", "uid"):
                np.testing.assert_array_equal(a[key], b[key], err_msg=ctx)
            for key in ("w", "cf1", "cf2", "cen", "pref"):
                assert np.array_equal(a[key], b[key]), (ctx, kind, key)
        got, exp = h.final_clusters, o.clusters
        assert [c.members_in_merge_order for c in got] == [[int(x) for x in c["members"]] for c in exp], ctx
        for g_, e_ in zip(got, exp):
            assert g_.cumulative_weight == e_["w"], ctx
            assert np.array_equal(g_.CF1, e_["cf1"]) and np.array_equal(g_.CF2, e_["cf2"]), ctx
            assert np.array_equal(g_.cluster_centroids, e_["cen"]), ctx
            assert np.array_equal(g_.preferred_dimension_vector, e_["pref"]), ctx

"""GPU parity tests: the HIP path (through the C-ABI) against the golden vectors of the upstream reference
and against the CPU oracle on seeded inputs.  Integer outputs (labels, ids, list order, merge order) must be
identical; float tables are compared bit for bit (the kernels keep the reference's operation order)."""
import os

import numpy as np
import pytest

import scenarios
from golden_util import GOLDEN, StateDump, assert_tables_equal, blob_inputs

pytestmark = pytest.mark.gpu


def _hdd(config, **tuning):
    from chronoclust_amd.clustering.hddstream import HDDStream
    return HDDStream(config, tuning=tuning or None)


def _check_against_dump(h, dump, t):
    par = dump.get(t, "params")
    assert (h.pi, h.mu, h.omicron) == (par[0], par[1], par[2])
    np.testing.assert_array_equal(h.labels_uid, dump.get(t, "labels_uid"), err_msg="labels t=%d" % t)
    assert (h.pcore_MC_last_id, h.outlier_MC_last_id) == (int(par[3]), int(par[4]))
    assert_tables_equal(h.table(0), dump, t, "pcore")
    assert_tables_equal(h.table(1), dump, t, "outlier")
    exp = dump.clusters(t)
    got = h.final_clusters
    assert len(got) == len(exp)
    for g, e in zip(got, exp):
        assert g.members_in_merge_order == [int(x) for x in e["members"]]
        assert g.cumulative_weight == e["w"]
        for a, b in ((g.CF1, e["cf1"]), (g.CF2, e["cf2"]), (g.cluster_centroids, e["cen"]),
                     (g.preferred_dimension_vector, e["pref"])):
            assert np.array_equal(a, b)


def _replay_dump(dump, Xs, config, **tuning):
    h = _hdd(config, **tuning)
    for t in range(dump.n_timepoints):
        h.online_microcluster_maintenance(Xs[t], int(dump.get(t, "daystamp")))
        _check_against_dump(h, dump, t)
    return h


@pytest.mark.parametrize("window", [1024, 64, 1])
def test_c1_golden(window):
    dump = StateDump(os.path.join(GOLDEN, "c1", "hdd_state.npz"))
    Xs = [dump.get(t, "X") for t in range(dump.n_timepoints)]
    if window == 1:
        Xs = [x[:600] for x in Xs]  # one point per window: only the first timepoint prefix is comparable
        h = _hdd(scenarios.params_to_config(scenarios.C1_PARAMS), window=1)
        from oracle import oracle as O
        o = O.OracleHDDStream(scenarios.params_to_config(scenarios.C1_PARAMS))
        for t, x in enumerate(Xs):
            h.online_microcluster_maintenance(x, t)
            o.online_microcluster_maintenance(x, t)
            _check_against_oracle(h, o)
        return
    _replay_dump(dump, Xs, scenarios.params_to_config(scenarios.C1_PARAMS), window=window)


def test_nocluster_golden():
    dump = StateDump(os.path.join(GOLDEN, "nocluster", "hdd_state.npz"))
    Xs = [dump.get(t, "X") for t in range(dump.n_timepoints)]
    _replay_dump(dump, Xs, scenarios.params_to_config(scenarios.NOCLUSTER_PARAMS))


@pytest.mark.parametrize("name", sorted(scenarios.BLOB_SCENARIOS))
def test_blob_golden(name):
    dump = StateDump(os.path.join(GOLDEN, "blob_%s.npz" % name))
    Xs = blob_inputs(name, dump)
    _replay_dump(dump, Xs, scenarios.params_to_config(scenarios.BLOB_SCENARIOS[name]["params"]))


def _check_against_oracle(h, o):
    from oracle import oracle as O
    np.testing.assert_array_equal(h.labels_uid, o.labels_uid)
    np.testing.assert_array_equal(h.labels_path, o.paths)
    assert (h.pcore_MC_last_id, h.outlier_MC_last_id) == o.counters
    for kind in (0, 1):
        a, b = h.table(kind), o.table(kind)
        for key in ("id", "uid"):
            np.testing.assert_array_equal(a[key], b[key])
        for key in ("w", "cf1", "cf2", "cen", "pref"):
            assert np.array_equal(a[key], b[key]), (kind, key)
    got, exp = h.final_clusters, o.clusters
    assert len(got) == len(exp)
    for g, e in zip(got, exp):
        assert g.members_in_merge_order == [int(x) for x in e["members"]]
        assert g.cumulative_weight == e["w"]
        assert np.array_equal(g.CF1, e["cf1"]) and np.array_equal(g.CF2, e["cf2"])
        assert np.array_equal(g.cluster_centroids, e["cen"])
        assert np.array_equal(g.preferred_dimension_vector, e["pref"])


SEEDED = [
    # (seed, n, d, g, sigma, params-overrides, window)
    (1, 6000, 20, 300, 0.01, {}, 1024),
    (2, 6000, 20, 300, 0.01, {}, 256),
    (3, 4000, 3, 6, 0.05, dict(param_epsilon=0.08, param_k=4, param_pi=3), 512),       # heavy overlap, tiny M
    (4, 4000, 14, 50, 0.02, dict(param_k=3, param_pi=9, param_epsilon=0.12), 1024),     # division path + filter
    (5, 3000, 40, 40, 0.01, {}, 1024),
    (6, 3000, 64, 20, 0.01, {}, 512),                                                   # CC_MAX_DIM
    (7, 5000, 8, 30, 0.03, dict(param_k=1), 1024),                                      # API default k = 1
    (8, 2500, 5, 10, 0.2, dict(param_epsilon=0.05), 1024),                              # almost everything is noise
    (9, 3000, 4, 15, 0.02, dict(param_k=2), 512),       # the remaining widths k_scan_u is compiled for: 4, 16, 32
    (10, 3000, 16, 40, 0.01, dict(param_k=8), 1024),
    (11, 3000, 32, 30, 0.01, {}, 512),
]


@pytest.mark.parametrize("seed,n,d,g,sigma,over,window", SEEDED)
def test_seeded_against_oracle(seed, n, d, g, sigma, over, window):
    from oracle import oracle as O
    params = scenarios.blob_params(n, **over)
    cfg = scenarios.params_to_config(params)
    h = _hdd(cfg, window=window, lookahead=3 if seed % 2 else 2)  # odd seeds: lookahead scans from the first window on
    o = O.OracleHDDStream(cfg)
    rng = np.random.default_rng(seed)
    for t in range(3):
        X = scenarios.make_blobs(seed * 100 + t, n, d, g, sigma)
        if t == 2:
            X = X[rng.permutation(n)[: n // 2]]  # ragged timepoint sizes
        h.online_microcluster_maintenance(X, t)
        o.online_microcluster_maintenance(X, t)
        _check_against_oracle(h, o)
    s = h.stats()
    assert s["points"] == len(X)
    # which snapshot scan ran: rows as scalar operands (k_scan_u) exactly when k is a power of two, the pdim filter is
    # off and d is one of the compiled widths; the LDS-staged k_scan otherwise (division path, filter, padded d)
    k = float(cfg["k"])
    pow2 = k > 0 and np.log2(k) == round(np.log2(k))
    filter_on = 0 < float(cfg["pi"]) < d
    expect_u = pow2 and not filter_on and d in (4, 8, 14, 16, 20, 32, 40, 64)
    if s["windows"] > 0:  # (a small table may have gone to the sequential kernel altogether: no scans at all)
        assert (s["scan_u_launches"] > 0) == expect_u, (s["scan_u_launches"], k, cfg["pi"], d)


@pytest.mark.parametrize("wps,window,sigma,eps", [(1, 64, 0.2, 0.05), (1, 96, 0.08, 0.06), (3, 64, 0.2, 0.05)])
def test_lookahead_across_batches_and_table_growth(wps, window, sigma, eps):
    """Lookahead windows at batch boundaries (host read-back after every `wps` windows) on overlapping, noisy data
    whose table outgrows its allocation several times: the carry marks of the last commit have to survive a
    reallocation, or stale snapshot candidates are taken for clean ones."""
    from oracle import oracle as O
    n, d, g = 7000, 5, 10
    cfg = scenarios.params_to_config(scenarios.blob_params(n, param_epsilon=eps))
    h = _hdd(cfg, window=window, windows_per_sync=wps, lookahead=3)
    o = O.OracleHDDStream(cfg)
    for t in range(2):
        X = scenarios.make_blobs(800 + t, n, d, g, sigma)
        h.online_microcluster_maintenance(X, t)
        o.online_microcluster_maintenance(X, t)
        _check_against_oracle(h, o)
    assert len(o.table(1)["id"]) + len(o.table(0)["id"]) > 1100  # the table was reallocated at least once


@pytest.mark.parametrize("d", [6, 8, 14])  # 6: padded to 8 columns (k_scan); 8, 14: rows as scalar operands (k_scan_u)
@pytest.mark.parametrize("scale,k,all_dims", [(1e-154, 4.0, False), (1e-158, 2.0, False), (1e-200, 4.0, False),
                                              (3e-121, 16.0, False), (1e-153, 4.0, True), (2e-154, 2.0, True)])
def test_subnormal_distance_terms(scale, k, all_dims, d):
    """The scans fuse `acc + x2 / pref` into one fma when k is a power of two, which is only the same double while
    x2 / pref stays a normal number; waves / tiles that hold a nonzero coordinate below 2^-400 must take the
    unfused path (CC_TINY in cc_online.h).  Some dimensions are scaled so that squared differences are subnormal
    (or flush to zero), others stay ordinary; at scale 3e-121 the tiny coordinates sit just above the threshold.
    all_dims: every coordinate (and epsilon) is scaled, so whole distances and radii are subnormal."""
    from oracle import oracle as O
    n, g = 3000, 12
    cfg = scenarios.params_to_config(scenarios.blob_params(n, param_k=k, param_epsilon=0.06 * (scale if all_dims else 1.0),
                                                           param_delta=0.05 * (scale if all_dims else 1.0)))
    h, o = _hdd(cfg, window=512, lookahead=3), O.OracleHDDStream(cfg)
    for t in range(2):
        X = scenarios.make_blobs(4100 + t, n, d, g, 0.02)
        if all_dims:
            X *= scale
        else:
            X[:, 1] *= scale
            X[:, 4] *= scale * 7.0
        X[::5, 2] = 0.0
        h.online_microcluster_maintenance(X, t)
        o.online_microcluster_maintenance(X, t)
        _check_against_oracle(h, o)


def test_continue_same_daystamp_and_edge_sizes():
    """hddstream.py:199: the same daystamp continues the timepoint (no decay); N = 1 and N = 0 inputs."""
    from oracle import oracle as O
    cfg = {"beta": 0.5, "delta": 0.3, "epsilon": 10, "lambda": 1, "k": 40, "mu": 1, "pi": 2, "omicron": 1,
           "upsilon": 1}
    h, o = _hdd(cfg), O.OracleHDDStream(cfg)
    first = np.array([[0.966970507, 0.185628831, 0.861853663], [0.557335192, 0.324320201, 0.495929691],
                      [0.698145385, 0.222617485, 0.83843284], [0.466592479, 0.557335192, 0.993609292]])
    for X in (first, np.array([[0.91259336, 0.16408931, 0.06039347]])):  # unittest_hddstream.py:43-88
        h.online_microcluster_maintenance(X, 0)
        o.online_microcluster_maintenance(X, 0)
        _check_against_oracle(h, o)
    assert h.table(0)["pref"][0].tolist() == [40, 40, 1]


def test_nonfinite_input_is_rejected():
    h = _hdd(scenarios.params_to_config(scenarios.blob_params(10)))
    X = np.random.rand(10, 4)
    X[3, 2] = np.nan
    with pytest.raises(ValueError):
        h.online_microcluster_maintenance(X, 0)


def test_assoc_argmin_against_oracle():
    from chronoclust_amd import _lib
    from oracle import oracle as O
    rng = np.random.default_rng(0)
    hd = _lib.Handle(0)
    for mc, mp, d, k in ((1, 1, 3, 4.0), (37, 129, 20, 4.0), (500, 700, 14, 3.0), (64, 64, 40, 1.0)):
        hd.set_params(0.01, 0.01, k, 0.5, 1.0, 0.0, 0.1, 0.01, 0.1, d)
        cur = rng.random((mc, d))
        pref = np.where(rng.random((mc, d)) < 0.5, k, 1.0)
        prev = rng.random((mp, d))
        prev[mp // 2] = prev[0]  # an exact tie: the first one must win
        gi, gd = hd.assoc_argmin(cur, pref, prev)
        oi, od = O.assoc_argmin(cur, pref, prev)
        np.testing.assert_array_equal(gi, oi)
        assert np.array_equal(gd, od)


def test_offline_intermediates_against_oracle():
    """Core flags, |N_eps|, PreDeCon pdim and |N_w| per pcore (predecon.py:136-217) on a scenario where
    neighbourhoods are non-trivial (upsilon large, anisotropic blobs, pi < d, k = 3)."""
    from oracle import oracle as O
    sc = scenarios.BLOB_SCENARIOS["d14_filter"]
    cfg = scenarios.params_to_config(sc["params"])
    Xs = scenarios.make_blob_timepoints(sc, raw=True)
    h, o = _hdd(cfg), O.OracleHDDStream(cfg)
    for t, X in enumerate(Xs):
        h.online_microcluster_maintenance(X, t)
        o.online_microcluster_maintenance(X, t)
        _, info = h._h.offline(dumps=True)
        for key in ("core", "pdim", "nn", "nw"):
            np.testing.assert_array_equal(info[key], o.offline_dump[key], err_msg="%s t=%d" % (key, t))
        assert info["nn"].max() > 1  # the eps-neighbourhoods are not all singletons


@pytest.mark.parametrize("n,d", [(1, 1), (7, 3), (1000, 20), (100003, 14), (5000, 64)])
def test_device_scaler_matches_the_restated_minmaxscaler(n, d):
    """cc_col_minmax / cc_points_upload_scaled / cc_points_download against chronoclust_amd.scaling.scaler.Scaler's
    numpy arithmetic (itself checked bit for bit against scikit-learn in tests/test_host_logic.py)."""
    from chronoclust_amd import _lib
    from chronoclust_amd.scaling.scaler import Scaler
    rng = np.random.default_rng(n + d)
    X = rng.normal(3.0, 50.0, (n, d)) * rng.uniform(1e-3, 1e3, d)
    if d > 2:
        X[:, 1] = 7.25          # zero range: scale_ stays 1
    if n > 5:
        X[3, 0] = X[:, 0].max() + 1.0
    hd = _lib.Handle(0)
    mn, mx = hd.col_minmax(X)
    np.testing.assert_array_equal(mn, X.min(axis=0))
    np.testing.assert_array_equal(mx, X.max(axis=0))
    Xn = X.copy()
    if n > 5:
        Xn[2, d - 1] = np.nan    # ignored by the fit, like np.nanmin / np.nanmax
        mn2, mx2 = hd.col_minmax(Xn)
        np.testing.assert_array_equal(mn2, np.nanmin(Xn, axis=0))
        np.testing.assert_array_equal(mx2, np.nanmax(Xn, axis=0))
    ref = Scaler()
    ref.fit_scaler(X)
    hd.points_upload_scaled(X, ref.scale_, ref.min_)
    scaled = hd.points_download(d)
    assert np.array_equal(scaled, ref.scale_data(X))
    back = hd.points_download(d, ref.scale_, ref.min_)
    assert np.array_equal(back, ref.reverse_scaling(ref.scale_data(X)))


@pytest.mark.parametrize("lookahead,wps,window", [(0, 4, 1024), (3, 4, 1024), (2, 2, 512), (0, 16, 2048)])
def test_quiet_stream_then_new_populations(lookahead, wps, window):
    """A long stretch in which no live version can matter to any point (the host stops launching the dirty scans and
    k_decide vouches for every point), then new populations appear in the middle of a batch: points that need the
    dirty scans are refused, the window commits up to them (possibly nothing: the device idles the batch) and the
    host brings the scans back.  Twice within one timepoint, then again after a decay."""
    from oracle import oracle as O
    d, g = 8, 30
    rng = np.random.default_rng(77)
    centres = rng.uniform(0.1, 0.9, (g + 6, d))

    def stretch(n, which, sigma=0.01):
        lab = rng.choice(which, n)
        return np.clip(centres[lab] + rng.normal(0.0, sigma, (n, d)), 0.0, 1.0)

    X0 = np.concatenate([stretch(26000, np.arange(g)),                       # quiet: 30 established populations
                         stretch(1500, np.arange(g, g + 3)),                  # three new ones, nothing else
                         stretch(9000, np.arange(g + 3)),                     # quiet again
                         stretch(2500, np.arange(g + 6))])                    # three more, mixed with the old
    X1 = np.concatenate([stretch(12000, np.arange(g + 6)), stretch(800, np.arange(3), sigma=0.2)])
    n = len(X0)
    cfg = scenarios.params_to_config(scenarios.blob_params(n, param_epsilon=0.06))
    h, o = _hdd(cfg, window=window, windows_per_sync=wps, lookahead=lookahead), O.OracleHDDStream(cfg)
    for t, X in enumerate((X0, X1)):
        h.online_microcluster_maintenance(X, t)
        o.online_microcluster_maintenance(X, t)
        _check_against_oracle(h, o)


@pytest.mark.parametrize("g,window,lookahead", [(8, 16384, 0), (3, 8192, 3), (40, 24576, 0), (150, 24576, 2),
                                                (150, 49152, 0), (40, 49152, 3), (300, 32768, 0)])  # 49 152: CC_MAX_WINDOW
def test_long_chains(g, window, lookahead):
    """Few microclusters and large windows: every MC absorbs hundreds to thousands of points per window, so the chains
    k_chain replays are far longer than the 32-entry member lists, and k_dseed finds live versions by reading the
    claims backwards (very long chains) or by walking the chain (chains a little over the list)."""
    from oracle import oracle as O
    n, d = 70000, 6
    cfg = scenarios.params_to_config(scenarios.blob_params(n, param_epsilon=0.08))
    h, o = _hdd(cfg, window=window, lookahead=lookahead), O.OracleHDDStream(cfg)
    for t in range(2):
        X = scenarios.make_blobs(5100 + t, n, d, g, 0.015)
        h.online_microcluster_maintenance(X, t)
        o.online_microcluster_maintenance(X, t)
        _check_against_oracle(h, o)


@pytest.mark.parametrize("window,lookahead,sigma", [(16384, 0, 0.048), (16384, 2, 0.048), (32768, 3, 0.046), (8192, 0, 0.049)])
def test_long_chains_with_rejected_steps(window, lookahead, sigma):
    """Few microclusters whose spread sits at the preferred-dimension threshold (delta = 0.05): their dimensions flip
    between preferred and not, so the radius test rejects steps all through the long chains of the pcore MCs.  Those
    chains are laid out ahead of k_chain (k_chain_long<.., true>: running sums only), every step evaluated by its own
    group, and a chain with a rejected step is replayed from that step on - both must have happened."""
    from oracle import oracle as O
    n, d, g = 70000, 6, 5
    cfg = scenarios.params_to_config(scenarios.blob_params(n, param_epsilon=0.08))
    h, o = _hdd(cfg, window=window, lookahead=lookahead, sequential=1), O.OracleHDDStream(cfg)  # (sequential = 1: never k_seq)
    tot = dict(long_prepared=0, long_replayed=0)
    for t in range(2):
        X = scenarios.make_blobs(5100 + t, n, d, g, sigma)
        h.online_microcluster_maintenance(X, t)
        o.online_microcluster_maintenance(X, t)
        _check_against_oracle(h, o)
        s = h.stats()
        for k in tot:
            tot[k] += s[k]
    if os.environ.get("CHRONOCLUST_HIP_LONGPREP") == "0" or os.environ.get("CHRONOCLUST_HIP_LONGCHAINS") == "0":
        assert tot["long_prepared"] == 0
    else:
        assert tot["long_prepared"] > 0 and 0 < tot["long_replayed"] < tot["long_prepared"], tot
    assert int(np.sum((h.labels_path & 3) == 1)) > 60  # (outlier adds: rejected by the pcore stage)


@pytest.mark.parametrize("window,lookahead,heavy", [(16384, 0, 0.3), (32768, 3, 0.5), (8192, 2, 0.1), (32768, 0, 0.02)])
def test_long_chains_on_a_large_table(window, lookahead, heavy):
    """Skewed populations on a table k_claims does not serve (more than 1 024 microclusters): three of 2 000 blobs take
    the share `heavy` of the events, so their chains run to thousands of members per window - listed by k_decide,
    replayed by k_chain_long from the second batch on (the first one still walks them in k_chain) - while the others
    stay on k_chain's listed path.  heavy = 0.02: chains just over the list's 32 entries."""
    from oracle import oracle as O
    n, d, g = 120_000, 6, 2000
    cfg = scenarios.params_to_config(scenarios.blob_params(n, param_epsilon=0.08))
    h, o = _hdd(cfg, window=window, lookahead=lookahead), O.OracleHDDStream(cfg)
    rng = np.random.default_rng(4242)
    centres = rng.uniform(0.1, 0.9, (g, d))
    for t in range(2):
        lab = rng.integers(3, g, n)
        big = rng.random(n) < heavy
        lab[big] = rng.integers(0, 3, int(big.sum()))
        X = np.ascontiguousarray(np.clip(centres[lab] + rng.normal(0.0, 0.004, (n, d)), 0.0, 1.0))
        h.online_microcluster_maintenance(X, t)
        o.online_microcluster_maintenance(X, t)
        _check_against_oracle(h, o)
    s = h.stats()
    assert s["rows"] > 1024
    assert s["long_chains"] > 0 and s["long_chain_launches"] > 0, s


@pytest.mark.parametrize("seed", range(12))
def test_skewed_streams_fuzz(seed):
    """Random skewed streams on tables of 1 100 - 2 600 microclusters: a few heavy populations (chains of hundreds to
    thousands of members, listed and replayed by k_chain_long from the second batch on) among many light ones, any
    window size up to the largest, lookahead off / default / forced, two timepoints with drift and decay."""
    from oracle import oracle as O
    rng = np.random.default_rng(8800 + seed)
    d = int(rng.choice([3, 6, 14, 20]))
    g = int(rng.integers(1100, 2600))
    n = int(rng.choice([40_000, 60_000]))
    heavy_blobs = int(rng.integers(1, 6))
    heavy = float(rng.choice([0.03, 0.1, 0.3, 0.6]))
    sigma = float(rng.choice([0.002, 0.004]))
    window = int(rng.choice([4096, 12288, 32768, 49152]))
    lookahead = int(rng.choice([0, 2, 3]))
    cfg = scenarios.params_to_config(scenarios.blob_params(
        n, param_epsilon=float(rng.choice([0.03, 0.08])), param_k=float(rng.choice([1.0, 2.0, 4.0])),
        param_lambda=float(rng.choice([0.0, 0.5])), promote_after=int(rng.choice([3, 10]))))
    h, o = _hdd(cfg, window=window, lookahead=lookahead), O.OracleHDDStream(cfg)
    centres = rng.uniform(0.05, 0.95, (g, d))
    for t in range(2):
        lab = rng.integers(heavy_blobs, g, n)
        big = rng.random(n) < heavy
        lab[big] = rng.integers(0, heavy_blobs, int(big.sum()))
        X = np.ascontiguousarray(np.clip(centres[lab] + rng.normal(0.0, sigma, (n, d)), 0.0, 1.0))
        h.online_microcluster_maintenance(X, t)
        o.online_microcluster_maintenance(X, t)
        _check_against_oracle(h, o)
        centres = np.clip(centres + rng.normal(0.0, 0.001, centres.shape), 0.0, 1.0)


@pytest.mark.parametrize("seed", range(16))
def test_long_chains_fuzz(seed):
    """Random streams of few microclusters (2 - 60 populations, every chain of a window hundreds to thousands of members
    long) at any compiled width - d <= 31: the pipelined layout of the running sums, one wave for all three; d >= 32: three
    waves, one batch after the other -, spreads from tight to the preferred-dimension threshold (rejected steps all through
    the chains: replays that start inside a chain), any window size, lookahead off / default / forced, two timepoints with
    drift and decay.  The sequential kernel is kept out of it (sequential = 1): it would take these streams over."""
    from oracle import oracle as O
    rng = np.random.default_rng(9900 + seed)
    d = int(rng.choice([3, 6, 14, 20, 31, 32, 40, 64]))
    g = int(rng.integers(2, 60))
    n = int(rng.choice([30_000, 50_000]))
    sigma = float(rng.choice([0.004, 0.015, 0.03, 0.046, 0.049]))
    window = int(rng.choice([4096, 16384, 32768, 49152]))
    lookahead = int(rng.choice([0, 2, 3]))
    k = float(rng.choice([1.0, 2.0, 3.0, 4.0]))
    # the radius threshold a little or well above the populations' own radius (d sigma^2 / k while every dimension is a
    # preferred one - below delta = 0.05 -, up to d sigma^2 when dimensions flip): absorbed, with rejections at the margin
    eps = float(np.sqrt(float(rng.choice([1.5, 4.0])) * d * sigma * sigma / k))
    cfg = scenarios.params_to_config(scenarios.blob_params(
        n, param_epsilon=eps, param_k=k, param_pi=int(rng.choice([0, max(1, d - 2)])),
        param_lambda=float(rng.choice([0.0, 0.5])), promote_after=int(rng.choice([3, 10]))))
    h, o = _hdd(cfg, window=window, lookahead=lookahead, sequential=1), O.OracleHDDStream(cfg)
    centres = rng.uniform(0.1, 0.9, (g, d))
    share = rng.dirichlet(np.full(g, 0.7))  # (uneven populations: a few take most of the events)
    for t in range(2):
        lab = rng.choice(g, n, p=share)
        X = np.ascontiguousarray(np.clip(centres[lab] + rng.normal(0.0, sigma, (n, d)), 0.0, 1.0))
        h.online_microcluster_maintenance(X, t)
        o.online_microcluster_maintenance(X, t)
        _check_against_oracle(h, o)
        centres = np.clip(centres + rng.normal(0.0, 0.002, centres.shape), 0.0, 1.0)


def test_windows_with_more_than_32767_creations():
    """k_commit_a ranks a window's creations and promotions in two 16-bit counts packed into one word: a window at
    the largest size (49 152 points) in which EVERY point creates a microcluster (uniform points, a radius threshold
    nothing passes) takes the creation count past 2^15 - ids, list order and labels must still be the reference's.
    (A call on a table of 1 024 rows or more opens with the configured window: the table comes from a saved state.)"""
    from oracle import oracle as O
    n0, n1, d = 1100, 52_000, 4
    rng = np.random.default_rng(99)
    X0 = np.ascontiguousarray(rng.uniform(0.0, 1.0, (n0, d)))
    X1 = np.ascontiguousarray(rng.uniform(0.0, 1.0, (n1, d)))
    cfg = scenarios.params_to_config(scenarios.blob_params(n1, param_epsilon=1e-7, param_k=2))
    o = O.OracleHDDStream(cfg)
    first = _hdd(cfg)
    first.online_microcluster_maintenance(X0, 0)
    o.online_microcluster_maintenance(X0, 0)
    _check_against_oracle(first, o)
    h = _hdd(cfg, window=49152, early_window=49152)
    h.set_state(first.get_state())
    h.online_microcluster_maintenance(X1, 0)  # (the same daystamp: no decay in between)
    o.online_microcluster_maintenance(X1, 0)
    _check_against_oracle(h, o)
    s = h.stats()
    assert h.outlier_MC_last_id == n0 + n1  # every point created one
    assert s["windows"] <= 2, s  # 49 152 + 2 848


def test_prefetched_upload_changes_nothing():
    """cc_points_prefetch: the next timepoint uploaded by a worker thread through page-locked staging while the current
    one is processed.  Same labels / tables as plain uploads; a prefetch of other data is discarded; NaN in
    prefetched data is reported by the upload that adopts it."""
    sc = scenarios.BLOB_SCENARIOS["d20"]
    cfg = scenarios.params_to_config(sc["params"])
    Xs = scenarios.make_blob_timepoints(sc, raw=True)
    plain, ahead = _hdd(cfg), _hdd(cfg)
    held = ahead.prefetch(Xs[0])
    for t in range(len(Xs)):
        plain.online_microcluster_maintenance(Xs[t], t)
        ahead.online_microcluster_maintenance(held, t)
        if t + 1 < len(Xs):
            held = ahead.prefetch(Xs[t + 1])
            if t == 1:
                ahead.prefetch(np.ascontiguousarray(Xs[0][:100]))  # replaced by another prefetch, never adopted
                held = Xs[t + 1]                                     # plain upload of the right data
        assert np.array_equal(plain.labels_uid, ahead.labels_uid)
        for kind in (0, 1):
            for key in ("id", "uid", "w", "cf1", "cf2", "cen", "pref"):
                assert np.array_equal(plain.table(kind)[key], ahead.table(kind)[key])
    bad = Xs[0].copy()
    bad[17, 3] = np.nan
    held = ahead.prefetch(bad)
    with pytest.raises(ValueError):
        ahead.online_microcluster_maintenance(held, 9)
    scaled = ahead.prefetch(Xs[0] * 3.0, device_scaling=(np.full(20, 1 / 3.0), np.zeros(20)))
    ahead._h.reset()
    ahead.last_data_timestamp = 0
    ahead.online_microcluster_maintenance(scaled, 0, device_scaling=(np.full(20, 1 / 3.0), np.zeros(20)))
    ref = _hdd(cfg)
    ref.online_microcluster_maintenance(Xs[0] * 3.0, 0, device_scaling=(np.full(20, 1 / 3.0), np.zeros(20)))
    assert np.array_equal(ahead.labels_uid, ref.labels_uid)


def test_point_clusters_device_gather_equals_numpy_join_and_reference_labels():
    """cc_point_clusters (per-point label -> creation number -> cluster, on the device) against the same join in numpy
    (chronoclust_amd.multi.point_cluster_index) on every timepoint of a churn scenario and of the bundled data, where
    outlier microclusters, pcores outside every cluster and merged clusters all occur."""
    from chronoclust_amd import multi
    seen_none = seen_cluster = False
    for name, cfg, Xs in (("d20", scenarios.params_to_config(scenarios.BLOB_SCENARIOS["d20"]["params"]),
                           scenarios.make_blob_timepoints(scenarios.BLOB_SCENARIOS["d20"])),
                          ("c1", scenarios.params_to_config(scenarios.C1_PARAMS), None)):
        if Xs is None:
            dump = StateDump(os.path.join(GOLDEN, "c1", "hdd_state.npz"))
            Xs = [dump.get(t, "X") for t in range(dump.n_timepoints)]
        h = _hdd(cfg)
        for t, X in enumerate(Xs):
            h.online_microcluster_maintenance(X, t)
            got = h.point_cluster_index()
            mem, off, *_ = h._cl_arrays
            pc = h.table(0)
            exp = multi.point_cluster_index(h.labels_uid, pc["id"], pc["uid"], mem, off)
            assert got.dtype == np.int64 and np.array_equal(got, exp), (name, t)
            seen_none |= bool((got < 0).any())
            seen_cluster |= bool((got >= 0).any())
    assert seen_none and seen_cluster


def test_heavy_rows_are_gathered_by_their_own_kernel():
    """Round 4: a microcluster whose chain was long is marked heavy; from the next batch on its claimants skip k_decide's
    three atomics and k_claims_heavy gathers first / last claimant, count and members.  (a) three populations take 30 %
    of the events, then the stream turns uniform (the rows are dropped from the list again); (b) every one of 1 200
    microclusters has a long chain (41 claimants per 49 152-point window): more nominations than the list of 64 holds.
    The oracle's results, with the kernel really used; and the same with CHRONOCLUST_HIP_HEAVY=0."""
    from oracle import oracle as O
    rng = np.random.default_rng(515)
    for case in ("skewed then uniform", "all chains long"):
        if case == "skewed then uniform":
            n, d, g, window = 300_000, 6, 1500, 8192
        else:
            n, d, g, window = 400_000, 6, 1200, 49152
        centres = rng.uniform(0.05, 0.95, (g, d))
        lab = rng.integers(0, g, n)
        if case == "skewed then uniform":
            big = (rng.random(n) < 0.3) & (np.arange(n) < n // 2)
            lab[big] = rng.integers(0, 3, int(big.sum()))
        X = np.ascontiguousarray(np.clip(centres[lab] + rng.normal(0.0, 0.003, (n, d)), 0.0, 1.0))
        cfg = scenarios.params_to_config(scenarios.blob_params(n, param_epsilon=0.05, promote_after=5))
        o = O.OracleHDDStream(cfg)
        o.online_microcluster_maintenance(X, 0)
        for knob in (None, "0"):
            old = os.environ.get("CHRONOCLUST_HIP_HEAVY")
            if knob is not None:
                os.environ["CHRONOCLUST_HIP_HEAVY"] = knob
            try:
                h = _hdd(cfg, window=window, lookahead=0)
            finally:
                if knob is not None:
                    if old is None:
                        os.environ.pop("CHRONOCLUST_HIP_HEAVY")
                    else:
                        os.environ["CHRONOCLUST_HIP_HEAVY"] = old
            h.online_microcluster_maintenance(X, 0)
            _check_against_oracle(h, o)
            s = h.stats()
            assert s["rows"] > 1024 and s["long_chains"] > 0
            assert (s["heavy_launches"] > 0) == (knob is None), (case, knob, s)

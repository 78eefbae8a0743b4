"""More than 64 dimensions (CC_WINDOW_MAX_DIM < d <= CC_MAX_DIM = 128; hddstream.py:107-114 takes any d): the online phase
runs on the sequential workgroup kernel (k_seq_g, the reference's loop on the table in HBM) from the first point on, the
offline phase and the trackers on their d = 128 instantiations - against the oracle like every other path: labels, both
tables bit for bit, clusters in merge order, tracking."""
import numpy as np
import pytest

import scenarios
from test_hip_parity import _check_against_oracle, _hdd

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(10))
def test_wide_streams_fuzz(seed):
    from oracle import oracle as O
    rng = np.random.default_rng(6500 + seed)
    d = int(rng.choice([65, 72, 96, 100, 127, 128]))
    g = int(rng.integers(2, 40))
    n = int(rng.choice([1500, 4000]))
    sigma = float(rng.choice([0.004, 0.015, 0.03, 0.049]))
    k = float(rng.choice([1.0, 2.0, 3.0, 4.0]))
    eps = float(np.sqrt(float(rng.choice([1.5, 4.0])) * d * sigma * sigma / k))
    cfg = scenarios.params_to_config(scenarios.blob_params(
        n, param_epsilon=eps, param_k=k, param_pi=int(rng.choice([0, d - 3])), param_lambda=float(rng.choice([0.0, 0.5, 2.0])),
        param_omicron=float(rng.choice([0.0, 0.0003])), promote_after=int(rng.choice([3, 10]))))
    h, o = _hdd(cfg), O.OracleHDDStream(cfg)
    centres = rng.uniform(0.1, 0.9, (g, d))
    for t in range(3):
        X = np.ascontiguousarray(np.clip(centres[rng.integers(0, g, n)] + rng.normal(0.0, sigma, (n, d)), 0.0, 1.0))
        h.online_microcluster_maintenance(X, t)
        o.online_microcluster_maintenance(X, t)
        _check_against_oracle(h, o)
        s = h.stats()
        assert s["seq_g_points"] == n and s["windows"] == 0  # (every point on k_seq_g, no window)
        centres = np.clip(centres + rng.normal(0.0, 0.004, centres.shape), 0.0, 1.0)
        if t == 1:
            centres = centres[: max(1, g - 2)]  # (two populations end: decay, downgrade, deletion)
            g = len(centres)


def test_wide_points_are_refused_in_a_group():
    """The windowed path is the only one the ranks of a group can share, and it stops at 64 dimensions: the call fails
    (before any collective) instead of clustering such points some other way."""
    from chronoclust_amd import _lib
    from chronoclust_amd.clustering.hddstream import HDDStream
    n, d = 500, 80
    cfg = scenarios.params_to_config(scenarios.blob_params(n))
    s, peer = HDDStream(cfg), HDDStream(cfg)
    _lib.comm_init_local([s._h, peer._h])
    with pytest.raises(ValueError, match="dimensions"):  # (CC_ERR_BAD_ARG maps to ValueError)
        s.online_microcluster_maintenance(scenarios.make_blobs(3, n, d, 5), 0)  # (fails before its first collective: no peer needed)


def test_wide_assoc_argmin_against_oracle():
    """TrackByHistoricalAssociation's nearest previous cluster (cluster_tracker.py:120-144) at d = 65 .. 128: the d = 128
    instantiation of k_assoc_tiled, k a power of two, not one, and 1 (the unit-operand form)."""
    from chronoclust_amd import _lib
    from oracle import oracle as O
    rng = np.random.default_rng(1)
    hd = _lib.Handle(0)
    for mc, mp, d, k in ((1, 1, 65, 4.0), (37, 129, 100, 4.0), (300, 500, 128, 3.0), (64, 64, 96, 1.0)):
        hd.set_params(0.01, 0.01, k, 0.5, 1.0, 0.0, 0.1, 0.01, 0.1, d)
        cur = rng.random((mc, d))
        pref = np.where(rng.random((mc, d)) < 0.5, k, 1.0)
        prev = rng.random((mp, d))
        prev[mp // 2] = prev[0]  # an exact tie: the first one must win
        gi, gd = hd.assoc_argmin(cur, pref, prev)
        oi, od = O.assoc_argmin(cur, pref, prev)
        np.testing.assert_array_equal(gi, oi)
        assert np.array_equal(gd, od)


def test_wide_offline_intermediates_against_oracle():
    """Core flags, |N_eps|, PreDeCon pdim and |N_w| per pcore (predecon.py:136-217) at d = 100 with neighbourhoods that are
    not trivial (upsilon large, anisotropic blobs, pi < d, k = 3): the d = 128 instantiation of k_eps_neighbours."""
    from oracle import oracle as O
    d, n, g = 100, 3000, 30
    rng = np.random.default_rng(2)
    cfg = scenarios.params_to_config(scenarios.blob_params(n, param_epsilon=0.6, param_pi=d - 10, param_k=3, param_upsilon=9.0,
                                                            param_omicron=0.0002, param_lambda=1.5))
    h, o = _hdd(cfg), O.OracleHDDStream(cfg)
    centres = rng.uniform(0.2, 0.8, (g, d))
    wide = rng.random((g, d)) < 0.1
    for t in range(2):
        lab = rng.integers(0, g, n)
        X = np.ascontiguousarray(np.clip(centres[lab] + rng.normal(0.0, 1.0, (n, d)) * np.where(wide[lab], 0.08, 0.01), 0.0, 1.0))
        h.online_microcluster_maintenance(X, t)
        o.online_microcluster_maintenance(X, t)
        _check_against_oracle(h, o)
        _, info = h._h.offline(dumps=True)
        for key in ("core", "pdim", "nn", "nw"):
            np.testing.assert_array_equal(info[key], o.offline_dump[key], err_msg="%s t=%d" % (key, t))
    assert len(h.final_clusters) > 0 and int(np.max(o.offline_dump["nn"])) > 1

"""The pruned snapshot scan (k_seed -> k_seed_merge -> k_scan_p) against the oracle and against the
plain scan, with the pruning FORCED for every window (CHRONOCLUST_HIP_PRUNE=2) - also where the library's own policy would
not use it (start-up, overlapping data): abandoned rows leave bounds instead of candidates, and every decision must
still be the reference's.  The knob is read when a handle is created, so each case sets it around the constructor."""
import os

import numpy as np
import pytest

import scenarios

pytestmark = pytest.mark.gpu


class _env(object):
    def __init__(self, **kv):
        self.kv = {k: str(v) for k, v in kv.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _hdd(cfg, prune, F=None, **tuning):
    from chronoclust_amd.clustering.hddstream import HDDStream
    kv = dict(CHRONOCLUST_HIP_PRUNE=prune)
    if F is not None:
        kv["CHRONOCLUST_HIP_PRUNE_F"] = F
    with _env(**kv):
        return HDDStream(cfg, tuning=tuning or None)


def _same_state(a, b):
    assert np.array_equal(a.labels_uid, b.labels_uid) and np.array_equal(a.labels_path, b.labels_path)
    assert (a.pcore_MC_last_id, a.outlier_MC_last_id) == (b.pcore_MC_last_id, b.outlier_MC_last_id)
    for kind in (0, 1):
        x, y = a.table(kind), b.table(kind)
        for key in ("id", "uid", "w", "cf1", "cf2", "cen", "pref"):
            assert np.array_equal(x[key], y[key]), (kind, key)
    assert [c.members_in_merge_order for c in a.final_clusters] == [c.members_in_merge_order for c in b.final_clusters]


def _against_oracle(h, o):
    assert np.array_equal(h.labels_uid, o.labels_uid) and np.array_equal(h.labels_path, o.paths)
    assert (h.pcore_MC_last_id, h.outlier_MC_last_id) == o.counters
    for kind in (0, 1):
        a, b = h.table(kind), o.table(kind)
        for key in ("id", "uid", "w", "cf1", "cf2", "cen", "pref"):
            assert np.array_equal(a[key], b[key]), (kind, key)
    assert [c.members_in_merge_order for c in h.final_clusters] == [[int(x) for x in c["members"]] for c in o.clusters]


CASES = [
    # (seed, n, d, g, sigma, parameter overrides, tuning, shift): the widths the pruned scan is compiled for
    (21, 9000, 20, 300, 0.01, {}, dict(window=2048), 0.0),
    (22, 9000, 14, 120, 0.01, {}, dict(window=1024, lookahead=3), 0.0),
    (23, 6000, 40, 150, 0.01, {}, dict(window=4096), 0.0),
    (24, 5000, 16, 60, 0.02, dict(param_k=8), dict(window=512, lookahead=2), 0.0),
    (25, 5000, 32, 60, 0.01, dict(param_k=2), dict(window=1024), 0.0),
    (26, 4000, 64, 30, 0.01, {}, dict(window=1024), 0.0),
    # k < 1 (1 / k > 1: a preferred dimension weighs MORE - the other branch of s_min / s_max in the T32 bound)
    (27, 6000, 20, 100, 0.01, dict(param_k=0.5), dict(window=1024), 0.0),
    # heavy overlap: nearly every row survives the prefix and is completed in phase B
    (28, 5000, 20, 12, 0.15, dict(param_epsilon=0.2), dict(window=512), 0.0),
    # coordinates around 1 000 (sigma 0.01): single precision resolves 6e-5 there - the bound must absorb it
    (29, 6000, 20, 200, 0.01, {}, dict(window=2048), 1000.0),
    (30, 6000, 14, 100, 0.01, {}, dict(window=1024), -250000.0),
]


@pytest.mark.parametrize("seed,n,d,g,sigma,over,tuning,shift", CASES)
def test_forced_pruning_matches_the_oracle(seed, n, d, g, sigma, over, tuning, shift):
    from oracle import oracle as O
    cfg = scenarios.params_to_config(scenarios.blob_params(n, **over))
    h = _hdd(cfg, 2, **tuning)
    o = O.OracleHDDStream(cfg)
    full = rows = launches = 0
    for t in range(3):
        X = scenarios.make_blobs(seed * 100 + t, n, d, g, sigma) + shift
        h.online_microcluster_maintenance(X, t)
        o.online_microcluster_maintenance(X, t)
        _against_oracle(h, o)
        s = h.stats()
        launches += s["scan_p_launches"]
        rows += s["pruned_scan_rows"]
        full += s["pruned_scan_full_rows"]
    assert launches > 0 and rows > 0  # the pruned kernels really ran
    print("seed %d d %d: %d pruned launches, %.1f %% of the sampled (wave, row) pairs completed" % (seed, d, launches, 100.0 * full / rows))


# Coordinates outside single precision's range (|x| >= 2^128: float(p) and float(c) are both +inf, their difference NaN)
# and far below it (squares underflow to 0): phase A must keep what it cannot judge.  `normalise_data=False` makes such
# float64 input legal (app.py:172-176), and the reference handles it like any other.
EXTREME = [
    # (name, factor applied to every coordinate, (dimension, factor) applied to one column or None)
    ("2^70", 2.0 ** 70, None),        # squares overflow single precision (2^140), coordinates do not
    ("2^130", 2.0 ** 130, None),      # coordinates themselves overflow single precision: inf - inf in the prefix
    ("2^-80", 2.0 ** -80, None),      # squares underflow single precision
    ("one huge prefix column", 1.0, (3, 2.0 ** 130)),
    ("one huge late column", 1.0, (13, 2.0 ** 130)),
]


@pytest.mark.parametrize("name,scale,column", EXTREME, ids=[e[0] for e in EXTREME])
@pytest.mark.parametrize("prune", [2, 1])
def test_coordinates_outside_single_precision_range(name, scale, column, prune):
    from oracle import oracle as O
    n, d, g = 5000, 20, 80
    eps = 0.05 * scale * (column[1] if column else 1.0)
    cfg = scenarios.params_to_config(scenarios.blob_params(n, param_epsilon=eps))
    h = _hdd(cfg, prune, window=1024)
    o = O.OracleHDDStream(cfg)
    launches = 0
    for t in range(2):
        X = scenarios.make_blobs(900 + t, n, d, g) * scale  # (a power of two: exact)
        if column:
            X[:, column[0]] *= column[1]
        assert np.isfinite(X).all()
        h.online_microcluster_maintenance(X, t)
        o.online_microcluster_maintenance(X, t)
        _against_oracle(h, o)
        launches += h.stats()["scan_p_launches"]
    if prune == 2:
        assert launches > 0


@pytest.mark.parametrize("F", [1, 2, 64, 4096])
def test_threshold_factor_never_changes_a_result(F):
    """F = 1 (thresholds equal to the seed's distance: bounds everywhere, the second-best almost never exact) up to
    F = 4096 (next to nothing abandoned): window after window the same labels and tables as the plain scan."""
    n, d, g = 12000, 20, 400
    cfg = scenarios.params_to_config(scenarios.blob_params(n, param_omicron=0.0002, param_lambda=2))
    sc = dict(seed=77, n=n, d=d, g=g, sigma=0.01, timepoints=3, drift=0.01, churn=0.08)
    Xs = scenarios.make_blob_timepoints(sc, raw=True)
    # (sequential = 1: at F = 1 so many points are refused that the policy would hand the stream to the sequential kernel -
    # k_seq_g, the table being beyond k_seq's image - and the last call would hold no pruned scan to speak of)
    plain = _hdd(cfg, 0, window=2048, sequential=1)
    pruned = _hdd(cfg, 2, F=F, window=2048, sequential=1)
    for t, X in enumerate(Xs):
        plain.online_microcluster_maintenance(X, t)
        pruned.online_microcluster_maintenance(X, t)
        _same_state(pruned, plain)
    assert pruned.stats()["scan_p_launches"] > 0 and plain.stats()["scan_p_launches"] == 0


def test_default_policy_uses_it_in_the_steady_state_and_not_while_microclusters_are_created():
    n, d, g = 400_000, 20, 2000
    X = scenarios.make_blobs(5, n, d, g)
    cfg = scenarios.params_to_config(scenarios.blob_params(n))
    auto, plain = _hdd(cfg, 1), _hdd(cfg, 0)
    auto.online_microcluster_maintenance(X, 0)
    plain.online_microcluster_maintenance(X, 0)
    _same_state(auto, plain)
    s = auto.stats()
    assert 0 < s["scan_p_launches"] < s["scan_u_launches"]  # (scan_u_launches counts both kinds of launch)
    # while the table filled plain scans ran; once it had settled the pruned chain came back (on a probe's word - 128 points
    # of a batch's first window - or at once when no tile needed its dirty scan), with guessed thresholds
    assert s["scan_g_launches"] > 0
    assert s["pruned_scan_full_rows"] < 0.5 * s["pruned_scan_rows"]  # (the sample includes the probes of the start-up phase)


def _fuzz_case(seed):
    """Random small streams inside the pruned scan's domain (k a power of two, no pdim filter, a compiled width > 8):
    duplicates and exact distance ties (grid data), sigma = 0, one-point timepoints, windows of any size."""
    rng = np.random.default_rng(7000 + seed)
    d = int(rng.choice([14, 16, 20, 32, 40, 64]))
    n = int(rng.choice([1, 17, 300, 1500, 4000]))
    g = int(rng.choice([1, 2, 5, 12, 40, 150]))
    sigma = float(rng.choice([0.0, 0.001, 0.02, 0.08, 0.3]))
    grid = bool(rng.random() < 0.3)
    cfg = {
        "beta": float(rng.choice([0.1, 0.5, 0.9, 1.0])),
        "delta": float(rng.choice([0.0, 0.01, 0.05, 0.3, 1.0])),
        "epsilon": float(rng.choice([0.001, 0.03, 0.1, 0.5, 3.0])),
        "lambda": float(rng.choice([0.0, 0.5, 2.0, 5.0])),
        "k": float(rng.choice([0.5, 1.0, 2.0, 4.0, 16.0])),
        "mu": float(rng.choice([0.0005, 0.002, 0.01, 0.1])),
        "pi": int(rng.choice([0, d, d + 3])),
        "omicron": float(rng.choice([0.0, 1e-5, 1e-3, 0.05])),
        "upsilon": float(rng.choice([0.5, 1.0, 3.0, 6.5, 20.0])),
    }
    window = int(rng.choice([5, 64, 700, 4096]))
    lookahead = int(rng.choice([0, 2, 3]))
    F = float(rng.choice([1.0, 2.0, 16.0, 16.0, 1024.0]))
    shift = float(rng.choice([0.0, 0.0, 100.0, -3.0e4]))
    centres = rng.uniform(0.1, 0.9, (g, d))
    Xs = []
    for t in range(3):
        nt = max(1, int(n * rng.choice([1.0, 0.5, 0.1]))) if t else n
        lab = rng.integers(0, g, nt)
        X = np.clip(centres[lab] + rng.normal(0.0, 1.0, (nt, d)) * sigma, 0.0, 1.0)
        if grid:
            X = np.round(X * 8) / 8
        Xs.append(np.ascontiguousarray(X + shift))
        centres = np.clip(centres + rng.normal(0, 0.02, centres.shape), 0, 1)
    return cfg, window, lookahead, F, Xs


def _general_case(seed):
    """The same random streams OUTSIDE the common case (round 6, k_scan_p3<GENERAL>): k not a power of two and / or the pdim
    filter of hddstream.py:317-321 on (pi < d), at the widths that kernel is compiled for."""
    cfg, window, lookahead, F, Xs = _fuzz_case(seed)
    rng = np.random.default_rng(9100 + seed)
    d = Xs[0].shape[1]
    if d == 64:
        d = int(rng.choice([14, 20, 40]))
        Xs = [np.ascontiguousarray(X[:, :d]) for X in Xs]
    which = seed % 3  # 0: any k, 1: the filter, 2: both
    if which != 1:
        cfg["k"] = float(rng.choice([3.0, 1.5, 10.0, 0.3, 7.25]))
    if which != 0:
        cfg["pi"] = int(rng.choice([1, 3, d // 2, d - 1]))
    return cfg, window, lookahead, F, Xs


@pytest.mark.parametrize("seed", range(72))
def test_forced_pruning_fuzz_with_the_pdim_filter_and_any_k(seed):
    from oracle import oracle as O
    cfg, window, lookahead, F, Xs = _general_case(seed)
    h = _hdd(cfg, 2, F=F, window=window, lookahead=lookahead)
    o = O.OracleHDDStream(cfg)
    pruned = plain_u = windows = 0
    for t, X in enumerate(Xs):
        h.online_microcluster_maintenance(X, t)
        o.online_microcluster_maintenance(X, t)
        _against_oracle(h, o)
        s = h.stats()
        pruned, plain_u, windows = pruned + s["scan_p_launches"], plain_u + s["scan_u_launches"], windows + s["windows"]
    # the pruned chain ran wherever windows were scanned at all (small tables go to the sequential kernels), and none of the
    # common case's kernels did
    assert plain_u == 0 and (pruned > 0 or windows == 0), (pruned, plain_u, windows)
    if seed == 0:
        assert pruned > 0


@pytest.mark.parametrize("seed", range(96))
def test_forced_pruning_fuzz(seed):
    from oracle import oracle as O
    cfg, window, lookahead, F, Xs = _fuzz_case(seed)
    h = _hdd(cfg, 2, F=F, window=window, lookahead=lookahead)
    o = O.OracleHDDStream(cfg)
    for t, X in enumerate(Xs):
        h.online_microcluster_maintenance(X, t)
        o.online_microcluster_maintenance(X, t)
        _against_oracle(h, o)


def test_guessed_thresholds_miss_a_loose_population():
    """Guessed thresholds (F x the mean distance at which earlier points joined their microclusters) suit the tight
    populations of this stream and miss the loose ones: those points go through k_missed and the seeded chain - more of
    them per window than the list holds at times, the rest are refused and the windows commit short.  The oracle's
    results all the same, and the same results with the guesses switched off."""
    from oracle import oracle as O
    rng = np.random.default_rng(31)
    n, d, g = 60_000, 20, 300
    centres = rng.uniform(0.1, 0.9, (g, d))
    sig = np.where(np.arange(g) < 240, 0.004, 0.03)  # 60 loose populations: their points lie 50 x farther from their centroids
    cfg = scenarios.params_to_config(scenarios.blob_params(n, param_epsilon=0.25))
    o = O.OracleHDDStream(cfg)
    guess = _hdd(cfg, 2, window=8192)  # (pruning forced: the guesses are used as soon as a mean join distance exists)
    with _env(CHRONOCLUST_HIP_GUESS=0):
        seeded = _hdd(cfg, 2, window=8192)
    missed = guessed = 0
    for t in range(3):
        lab = rng.integers(0, g, n)
        X = np.ascontiguousarray(np.clip(centres[lab] + rng.normal(0.0, 1.0, (n, d)) * sig[lab, None], 0.0, 1.0))
        for h in (guess, seeded):
            h.online_microcluster_maintenance(X, t)
        o.online_microcluster_maintenance(X, t)
        _against_oracle(guess, o)
        _same_state(guess, seeded)
        s = guess.stats()
        missed += s["missed_points"]
        guessed += s["scan_g_launches"]
        assert seeded.stats()["scan_g_launches"] == 0
    assert guessed > 0 and missed > 0
    print("guessed-threshold launches %d, points missed %d" % (guessed, missed))


def test_lean_guessed_scans_and_a_population_that_appears_late():
    """Lean guessed scans (round 4): after a batch with guessed thresholds and no missed point the scans run without
    k_missed and the seeded chain.  A population that appears in the middle of a settled stream is then missed by the
    guess: its first point keeps a bound in first place, k_decide refuses it, the window commits up to it, the batch idles,
    and the policy - hearing of the refusal through stat_missed - brings the chain back.  The oracle's results all the
    same, with lean scans before and after; the same stream in a group of two ranks (no second all-gather while lean);
    and with CHRONOCLUST_HIP_LEAN=0."""
    from oracle import oracle as O
    rng = np.random.default_rng(77)
    n, d, g, late = 900_000, 20, 400, 12
    centres = rng.uniform(0.1, 0.9, (g + late, d))
    lab = rng.integers(0, g, n)
    for s in range(late):  # twelve populations that start at different places of the last third
        start = 2 * n // 3 + s * 20_000
        idx = start + np.flatnonzero(rng.random(n - start) < 0.004)
        lab[idx] = g + s
    X = np.ascontiguousarray(np.clip(centres[lab] + rng.normal(0.0, 0.006, (n, d)), 0.0, 1.0))
    cfg = scenarios.params_to_config(scenarios.blob_params(n))
    o = O.OracleHDDStream(cfg)
    o.online_microcluster_maintenance(X, 0)
    h = _hdd(cfg, 1, window=8192)  # (the library's own policy)
    h.online_microcluster_maintenance(X, 0)
    _against_oracle(h, o)
    s = h.stats()
    assert s["scan_lean_launches"] > 0 and s["scan_g_launches"] > s["scan_lean_launches"] and s["missed_points"] > 0, s
    with _env(CHRONOCLUST_HIP_LEAN=0):
        full = _hdd(cfg, 1, window=8192)
    full.online_microcluster_maintenance(X, 0)
    _same_state(h, full)
    assert full.stats()["scan_lean_launches"] == 0 and full.stats()["scan_g_launches"] > 0
    # split over two ranks (in-process group): the same results, lean windows without the second gather
    import pipeline_util as PU
    from test_sharded_local import run_group
    single = PU.run_pipeline([X], cfg, tuning=dict(window=8192))
    for r in run_group(2, [X], cfg, tuning=dict(window=8192)):
        PU.same_results(r, single)
        assert r[0]["stats"]["scan_lean_launches"] > 0 and r[0]["stats"]["sharded_windows"] > 0


def test_guessed_thresholds_and_points_whose_outlier_list_starts_with_a_bound():
    """A settled stream of 2 000 tight populations, then six WIDE ones (radius just over epsilon) appear late.  Once such a
    population's microcluster is promoted, many of its points find it within the guessed threshold (their pcore list is
    resolved: the point counts as found), fail its radius test - the MC is young and light - and go on to the outlier stage
    with no outlier MC within the guess: their outlier list starts with a bound and k_decide refuses them.  The window is cut
    short there, the next one starts at that point, and k_missed / k_missed_g put a window's first point on the seeded chain's
    list when the previous window was cut short at an undecidable point (Ctl::seed_at, round 5; ADVICE r04 traced a stall
    from the code: such a point refused again and again until the policy switched pruned scans off for the call).  On this
    stream the policy's other rules - back to seeded thresholds when one point in sixteen is missed - act first, with or
    without seed_at (profiles/r05_tool_late_wide.txt: identical traces), so what the test holds is exactness on a stream
    of this kind and that it ends on pruned scans."""
    from oracle import oracle as O
    rng = np.random.default_rng(2025)
    n, d, g, late = 600_000, 20, 2000, 6
    centres = rng.uniform(0.1, 0.9, (g + late, d))
    lab = rng.integers(0, g, n)
    for s in range(late):
        start = 2 * n // 3 + s * 12000
        idx = start + np.flatnonzero(rng.random(n - start) < 0.004)
        lab[idx] = g + s
    sig = np.where(lab >= g, 0.0225, 0.006)  # (d sigma^2 / k = 2.5e-3 = epsilon^2 for the late ones)
    X = np.ascontiguousarray(np.clip(centres[lab] + rng.normal(0.0, 1.0, (n, d)) * sig[:, None], 0.0, 1.0))
    cfg = scenarios.params_to_config(scenarios.blob_params(n))
    o = O.OracleHDDStream(cfg)
    o.online_microcluster_maintenance(X, 0)
    paths = np.asarray(o.paths)
    assert int((paths[2 * n // 3:] == 2).sum()) > 3 * late  # microclusters opened beside the late populations' own
    h = _hdd(cfg, 1, window=8192)  # (the library's own policy)
    h.online_microcluster_maintenance(X, 0)
    _against_oracle(h, o)
    s = h.stats()
    print("rows %d, windows %d truncated %d, scan launches %d of them pruned %d (guessed %d, lean %d), missed points %d" % (
        s["rows"], s["windows"], s["truncated"], s["scan_u_launches"], s["scan_p_launches"], s["scan_g_launches"],
        s["scan_lean_launches"], s["missed_points"]))
    assert s["scan_g_launches"] > 0 and s["missed_points"] > 0
    # the last third of the stream still runs on pruned scans (they would be ~0 there once switched off for the call)
    assert s["scan_p_launches"] >= s["scan_u_launches"] * 0.6, s


def test_forced_pruning_seed_whose_first_window_point_was_refused_three_batches_in_a_row():
    """Seed 9348 of the round-5 soak: with pruning forced on a table of eight microclusters a guessed threshold misses every
    point of a window; the batch that ran without dirty scans, the one with guessed thresholds and the one with seeded ones
    each commit nothing before the policy is down to plain scans - the call used to end with "no progress in three
    consecutive batches" (a liveness failure, not a wrong result).  The policy now ends pruned scans at the second such
    batch in a row and the limit is five."""
    test_forced_pruning_fuzz(9348)

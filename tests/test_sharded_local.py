"""The exact multi-GPU path verified on ONE GPU: `world` handles in one process form an
in-process group (cc_comm_init_local), each driven by its own host thread.  Every rank scans only its share of the
table rows, the ranks all-gather one candidate record per window point, and the offline / association pair
matrices are split by rows.  The claim under test: every rank ends with the results one GPU computes alone - bit for
bit (labels, tables, id counters, merge-ordered clusters, lineage and association strings) - and therefore with the
oracle's.  The RCCL transport is the same code path with ncclAllGather as the exchange; here it is exercised with a
communicator of one rank (all a single-GPU machine allows)."""
import threading

import numpy as np
import pytest

import pipeline_util as P
import scenarios

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


def run_group(world, Xs, cfg, tuning=None, min_row_dims=0, offline_min_rows=0, env=None, calibrate=False):
    """The pipeline of app.run on `world` replicas of one stream, one thread per rank.  Returns the per-timepoint
    results of every rank.  env: knobs the library reads when a handle is created (CHRONOCLUST_HIP_PRUNE ...)."""
    import os
    from chronoclust_amd import _lib
    from chronoclust_amd.clustering.hddstream import HDDStream
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update({k: str(v) for k, v in (env or {}).items()})
    try:
        streams = [HDDStream(cfg, tuning=tuning) for _ in range(world)]
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    _lib.comm_init_local([s._h for s in streams])
    for r, s in enumerate(streams):
        s._h.set_shard_thresholds(min_row_dims, offline_min_rows)
        assert s._h.comm_info() == dict(rank=r, world=world, transport="local")
    results, errors = [None] * world, [None] * world

    def work(rank):
        try:
            if calibrate:  # (collective: every member from its own thread; overrides the thresholds set above)
                streams[rank]._h.comm_calibrate()
            results[rank] = P.run_pipeline(Xs, cfg, stream=streams[rank])
        except BaseException as e:  # noqa: BLE001 - reported after the join
            errors[rank] = e
            try:
                streams[rank]._h.comm_destroy()  # the peers must not wait for this rank
            except Exception:
                pass

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in errors:
        if e is not None:
            raise e
    return results


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("name", ["d40", "d14_filter", "d20"])
def test_golden_scenarios_sharded_equal_single_and_reference(name, world, golden_dir):
    """Oracle-sized scenarios with every split forced on from the first window (thresholds 0): the ranks reproduce
    the single-GPU results, which the other parity tests pin to the Python reference's dumps."""
    sc = scenarios.BLOB_SCENARIOS[name]
    cfg = scenarios.params_to_config(sc["params"])
    Xs = scenarios.make_blob_timepoints(sc)
    single = P.run_pipeline(Xs, cfg)
    for tuning in (None, dict(window=512, segments=8, lookahead=3), dict(window=2048, lookahead=2)):
        for res in run_group(world, Xs, cfg, tuning=tuning):
            P.same_results(res, single)
            assert all(r["stats"]["sharded_windows"] == r["stats"]["windows"] for r in res)
    rec = np.load("%s/blob_%s.npz" % (golden_dir, name))
    for t, r in enumerate(single):
        assert np.array_equal(r["labels_uid"], rec["t%d_labels_uid" % t])


def test_sharded_matches_oracle_with_churn():
    from oracle import oracle as O
    sc = dict(seed=9, n=30_000, d=20, g=900, sigma=0.01, timepoints=3, drift=0.01, churn=0.08)
    cfg = scenarios.params_to_config(scenarios.blob_params(sc["n"], param_omicron=0.00007, param_lambda=2))
    Xs = scenarios.make_blob_timepoints(sc, raw=True)
    res = run_group(2, Xs, cfg)
    o = O.OracleHDDStream(cfg)
    for t, X in enumerate(Xs):
        o.online_microcluster_maintenance(X, t)
        for rank in range(2):
            r = res[rank][t]
            assert np.array_equal(r["labels_uid"], o.labels_uid)
            assert r["counters"] == o.counters
            for kind, name in ((0, "pcore"), (1, "outlier")):
                b = o.table(kind)
                for key in ("id", "uid", "w", "cf1", "cf2", "cen", "pref"):
                    assert np.array_equal(r[name][key], b[key]), (t, rank, kind, key)
            assert r["members"] == [[int(x) for x in c["members"]] for c in o.clusters]


def test_split_switches_on_while_the_table_grows():
    """Default-style threshold: the first windows run unsplit (small table), later ones split - the lookahead chain is
    restarted at the switch.  600 k x 20 with 25 000 blobs crosses rows * d = 400 000 while microclusters are created."""
    n, d, g = 600_000, 20, 25_000
    X = scenarios.make_blobs(7, n, d, g)
    cfg = scenarios.params_to_config(scenarios.blob_params(n))
    single = P.run_pipeline([X], cfg)
    res = run_group(2, [X], cfg, min_row_dims=400_000, offline_min_rows=8192)
    for r in res:
        P.same_results(r, single)
        st = r[0]["stats"]
        assert 0 < st["sharded_windows"] < st["windows"]


def test_c5_shaped_sharded_equals_single():
    """2 M x 40 with 50 000 microclusters (the stress config's table on one GPU's share of its points), two ranks
    with the default thresholds: scan split once the table holds 10 000 rows, offline pair matrices split."""
    n, d, g = 2_000_000, 40, 50_000
    X = scenarios.make_blobs(42, n, d, g)
    cfg = scenarios.params_to_config(scenarios.blob_params(n))
    single = P.run_pipeline([X], cfg)
    res = run_group(2, [X], cfg, min_row_dims=-1, offline_min_rows=-1)
    for r in res:
        P.same_results(r, single)
        assert r[0]["stats"]["sharded_windows"] > 0


def test_rccl_transport_with_one_rank():
    """ncclCommInitRank / ncclAllGather through the dlopen'ed librccl, communicator of one rank: the calls the
    multi-process path makes, on the streams it makes them on (scan forced through merge + all-gather)."""
    from chronoclust_amd import _lib
    from chronoclust_amd.clustering.hddstream import HDDStream
    sc = scenarios.BLOB_SCENARIOS["d20"]
    cfg = scenarios.params_to_config(sc["params"])
    Xs = scenarios.make_blob_timepoints(sc)
    single = P.run_pipeline(Xs, cfg)
    h = HDDStream(cfg)
    h._h.comm_init_rccl(_lib.comm_unique_id(), 0, 1)
    assert h._h.comm_info() == dict(rank=0, world=1, transport="rccl")
    h._h.set_shard_thresholds(0, 0)  # every scan, offline phase and association argmin goes through the exchange
    res = P.run_pipeline(Xs, cfg, stream=h)
    P.same_results(res, single)
    assert all(r["stats"]["sharded_windows"] == r["stats"]["windows"] > 0 for r in res)
    h._h.comm_destroy()
    assert h._h.comm_info()["transport"] == "none"


def test_eight_ranks_with_tiny_shares():
    """The rank count of the target node.  30 - 140 microclusters over eight ranks: shares of a handful of rows, empty
    shares at the start, offline blocks that are mostly padding - the same results as one GPU."""
    for name in ("d40", "d20"):
        sc = scenarios.BLOB_SCENARIOS[name]
        cfg = scenarios.params_to_config(sc["params"])
        Xs = scenarios.make_blob_timepoints(sc)
        single = P.run_pipeline(Xs, cfg)
        for res in run_group(8, Xs, cfg, tuning=dict(window=1024)):
            P.same_results(res, single)


def test_sharded_group_with_tiny_timepoints():
    """One-point and few-point timepoints between normal ones (windows of a single point, empty row shares): the ranks
    still agree with the single-GPU run."""
    rng = np.random.default_rng(4)
    centres = rng.uniform(0.2, 0.8, (6, 8))
    sizes = [1, 2, 1500, 5, 1, 800]
    Xs = [np.ascontiguousarray(np.clip(centres[rng.integers(0, 6, n)] + rng.normal(0, 0.01, (n, 8)), 0, 1)) for n in sizes]
    cfg = scenarios.params_to_config(scenarios.blob_params(1500, param_lambda=0.3, param_omicron=0.0005))
    single = P.run_pipeline(Xs, cfg)
    for res in run_group(3, Xs, cfg):
        P.same_results(res, single)


# ---------------------------------------------------------------------------------------------------------
# pruned snapshot scans split over the ranks (round 4): seeds and thresholds over all rows on every rank, phases A / B
# over the rank's rows, the rank's sample of completed rows gathered with its candidate records
# ---------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("seed", range(0, 96, 4))
def test_forced_pruning_fuzz_in_a_group(seed, world):
    """The forced-pruning fuzz of tests/test_pruned_scan.py with every scan split over 2 / 3 ranks: the oracle's
    results on every rank, and the pruned kernels really ran split."""
    from oracle import oracle as O
    from test_pruned_scan import _fuzz_case
    cfg, window, lookahead, F, Xs = _fuzz_case(seed)
    res = run_group(world, Xs, cfg, tuning=dict(window=window, lookahead=lookahead),
                    env=dict(CHRONOCLUST_HIP_PRUNE=2, CHRONOCLUST_HIP_PRUNE_F=F))
    o = O.OracleHDDStream(cfg)
    for t, X in enumerate(Xs):
        o.online_microcluster_maintenance(X, t)
        for rank in range(world):
            r = res[rank][t]
            assert np.array_equal(r["labels_uid"], o.labels_uid)
            assert r["counters"] == o.counters
            for kind, name in ((0, "pcore"), (1, "outlier")):
                b = o.table(kind)
                for key in ("id", "uid", "w", "cf1", "cf2", "cen", "pref"):
                    assert np.array_equal(r[name][key], b[key]), (t, rank, kind, key)
            assert r["members"] == [[int(x) for x in c["members"]] for c in o.clusters]
    st = [r["stats"] for r in res[0]]
    assert sum(s["scan_p_launches"] for s in st) > 0 and sum(s["sharded_windows"] for s in st) > 0


@pytest.mark.parametrize("world", [2, 3])
def test_pruned_and_split_in_the_steady_state(world):
    """The default policy inside a group whose table is above the split threshold: start-up with plain split scans, then
    pruned split scans - the counters the policy reads are the gathered ones, so the ranks keep deciding alike - and
    the single-GPU results."""
    n, d, g = 400_000, 20, 2000
    X = scenarios.make_blobs(5, n, d, g)
    cfg = scenarios.params_to_config(scenarios.blob_params(n))
    single = P.run_pipeline([X], cfg)
    res = run_group(world, [X], cfg, min_row_dims=20_000, offline_min_rows=-1)
    for r in res:
        P.same_results(r, single)
        st = r[0]["stats"]
        assert st["sharded_windows"] > 0 and 0 < st["scan_p_launches"] < st["scan_u_launches"]
        assert 0 < st["pruned_scan_full_rows"] < 0.2 * st["pruned_scan_rows"]
    assert len({(r[0]["stats"]["pruned_scan_rows"], r[0]["stats"]["pruned_scan_full_rows"], r[0]["stats"]["windows"]) for r in res}) == 1


@pytest.mark.parametrize("world", [2, 3])
def test_guessed_thresholds_in_a_group_miss_a_loose_population(world):
    """The stream of tests/test_pruned_scan.py::test_guessed_thresholds_miss_a_loose_population with every scan split:
    each rank scans its rows against the same guessed thresholds, the list of missed points is derived from the gathered
    records (the same list on every rank), the seeded chain runs for those points and their new records travel in a
    second, small all-gather - at times more points are missed than the list holds.  The single-GPU results on every
    rank, and the same counters everywhere (the ranks' policies must keep deciding alike)."""
    rng = np.random.default_rng(31)
    n, d, g = 60_000, 20, 300
    centres = rng.uniform(0.1, 0.9, (g, d))
    sig = np.where(np.arange(g) < 240, 0.004, 0.03)
    cfg = scenarios.params_to_config(scenarios.blob_params(n, param_epsilon=0.25))
    Xs = []
    for _ in range(3):
        lab = rng.integers(0, g, n)
        Xs.append(np.ascontiguousarray(np.clip(centres[lab] + rng.normal(0.0, 1.0, (n, d)) * sig[lab, None], 0.0, 1.0)))
    single = P.run_pipeline(Xs, cfg)
    res = run_group(world, Xs, cfg, tuning=dict(window=8192), env=dict(CHRONOCLUST_HIP_PRUNE=2))
    for r in res:
        P.same_results(r, single)
    per_rank = [[(s["stats"]["scan_g_launches"], s["stats"]["missed_points"], s["stats"]["windows"],
                  s["stats"]["sharded_windows"]) for s in r] for r in res]
    assert all(p == per_rank[0] for p in per_rank)
    assert sum(x[0] for x in per_rank[0]) > 0 and sum(x[1] for x in per_rank[0]) > 0
    assert all(x[2] == x[3] for x in per_rank[0])
    # and with the guesses switched off: seeds for every point, the same results
    off = run_group(world, Xs, cfg, tuning=dict(window=8192), env=dict(CHRONOCLUST_HIP_PRUNE=2, CHRONOCLUST_HIP_GUESS=0))
    for r in off:
        P.same_results(r, single)
        assert all(s["stats"]["scan_g_launches"] == 0 for s in r)


def test_rccl_transport_carries_the_second_gather_of_guessed_thresholds():
    """The second, fixed-size all-gather of a split scan with guessed thresholds (the missed points' new records) over
    ncclAllGather: a communicator of one rank that takes the group's steps on request (CHRONOCLUST_HIP_GROUP_GUESS=1) -
    k_missed_g over the gathered records, the seeded chain for the listed points, compact records, gather, scatter - on
    both streams.  The stream with loose populations (points are missed in every window), pruning forced."""
    import os
    from chronoclust_amd import _lib
    from chronoclust_amd.clustering.hddstream import HDDStream
    rng = np.random.default_rng(31)
    n, d, g = 60_000, 20, 300
    centres = rng.uniform(0.1, 0.9, (g, d))
    sig = np.where(np.arange(g) < 240, 0.004, 0.03)
    cfg = scenarios.params_to_config(scenarios.blob_params(n, param_epsilon=0.25))
    Xs = []
    for _ in range(2):
        lab = rng.integers(0, g, n)
        Xs.append(np.ascontiguousarray(np.clip(centres[lab] + rng.normal(0.0, 1.0, (n, d)) * sig[lab, None], 0.0, 1.0)))
    single = P.run_pipeline(Xs, cfg)
    env = dict(CHRONOCLUST_HIP_PRUNE="2", CHRONOCLUST_HIP_GROUP_GUESS="1")
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        h = HDDStream(cfg, tuning=dict(window=8192))
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    h._h.comm_init_rccl(_lib.comm_unique_id(), 0, 1)
    h._h.set_shard_thresholds(0, 0)
    res = P.run_pipeline(Xs, cfg, stream=h)
    P.same_results(res, single)
    assert sum(r["stats"]["scan_g_launches"] for r in res) > 0 and sum(r["stats"]["missed_points"] for r in res) > 0
    assert all(r["stats"]["sharded_windows"] == r["stats"]["windows"] > 0 for r in res)
    h._h.comm_destroy()


def test_skewed_stream_in_a_group():
    """Heavy rows (three populations take 30 % of the events) with every scan split over two ranks: the marks and the list
    of heavy rows are kept by the replicated validation kernels, the host's decision to launch k_claims_heavy rests on a
    counter that is the same on every rank - the single-GPU results, and the kernel really used."""
    rng = np.random.default_rng(99)
    n, d, g = 300_000, 14, 1500
    centres = rng.uniform(0.05, 0.95, (g, d))
    Xs = []
    for _ in range(2):
        lab = rng.integers(3, g, n)
        big = rng.random(n) < 0.3
        lab[big] = rng.integers(0, 3, int(big.sum()))
        Xs.append(np.ascontiguousarray(np.clip(centres[lab] + rng.normal(0.0, 0.004, (n, d)), 0.0, 1.0)))
    cfg = scenarios.params_to_config(scenarios.blob_params(n))
    single = P.run_pipeline(Xs, cfg, tuning=dict(window=8192))
    assert sum(r["stats"]["heavy_launches"] for r in single) > 0
    res = run_group(2, Xs, cfg, tuning=dict(window=8192))
    for r in res:
        P.same_results(r, single)
        assert sum(x["stats"]["heavy_launches"] for x in r) > 0 and all(x["stats"]["sharded_windows"] > 0 for x in r)
    assert [x["stats"]["heavy_launches"] for x in res[0]] == [x["stats"]["heavy_launches"] for x in res[1]]


@pytest.mark.parametrize("world", [2, 3])
def test_split_thresholds_come_from_a_measurement_and_agree_on_all_ranks(world):
    """cc_comm_calibrate: the all-gather of a window's records and a plain scan are TIMED when the group is formed, every rank
    takes the group's maxima and derives the same thresholds (plain scans from exchange x world / (world - 1) / scan per
    (row, dim) on, pruned chains from 3.3 times that); a stream clustered with those thresholds equals one GPU's."""
    n, d, g = 120_000, 20, 6000
    X = scenarios.make_blobs(11, n, d, g)
    cfg = scenarios.params_to_config(scenarios.blob_params(n))
    single = P.run_pipeline([X], cfg)
    res = run_group(world, [X], cfg, min_row_dims=-1, offline_min_rows=-1, calibrate=True)
    stats = [r[0]["stats"] for r in res]
    keys = ("calib_allgather_us", "calib_scan_ns_per_row_dim", "split_threshold_row_dims", "split_threshold_row_dims_pruned")
    assert all(tuple(s[k] for k in keys) == tuple(stats[0][k] for k in keys) for s in stats)  # the same on every rank
    s0 = stats[0]
    assert s0["calib_allgather_us"] > 1.0 and 0.3 < s0["calib_scan_ns_per_row_dim"] < 30.0
    want = s0["calib_allgather_us"] * 1e3 * world / (world - 1) / s0["calib_scan_ns_per_row_dim"]
    assert abs(s0["split_threshold_row_dims"] - want) <= 1.0
    assert abs(s0["split_threshold_row_dims_pruned"] - 3.3 * want) <= 4.0
    for r in res:
        P.same_results(r, single)
    print("world %d: all-gather %.1f us, scan %.2f ns per (row, dim): split from %d / %d row-dims on; %d of %d windows split" % (
        world, s0["calib_allgather_us"], s0["calib_scan_ns_per_row_dim"], s0["split_threshold_row_dims"],
        s0["split_threshold_row_dims_pruned"], s0["sharded_windows"], s0["windows"]))


def test_a_group_of_one_measures_but_keeps_its_thresholds():
    from chronoclust_amd import _lib
    h = _lib.Handle(0)
    h.comm_init_rccl(_lib.comm_unique_id(), 0, 1)  # (calibrates by itself)
    s = h.stats()
    assert s["calib_allgather_us"] > 0.0 and s["calib_scan_ns_per_row_dim"] > 0.0
    assert s["split_threshold_row_dims"] == 400_000 and s["split_threshold_row_dims_pruned"] == 400_000
    h.set_shard_thresholds(1234, -1)
    s = h.stats()
    assert s["split_threshold_row_dims"] == 1234 and s["split_threshold_row_dims_pruned"] == 1234
    h.close()

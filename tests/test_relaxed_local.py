"""The RELAXED multi-GPU mode (events of a timepoint sharded over the ranks, CF deltas all-reduced per super-step;
cc_comm_set_relaxed) on in-process groups of handles on one GPU.  It is not the reference's algorithm, so nothing here
is compared bit for bit with the reference; what is pinned:
  - every rank ends every timepoint with bit-identical tables, labels, counters and clusters (the property the
    replicated halves of the super-steps and the offline phase rely on);
  - conservation: every point is labelled, every MC's weight is the number of its points (first timepoint), CF1 is the
    sum of its points up to reassociation;
  - on well-separated blobs the partition of the points agrees with the exact path's;
  - a group of one rank over RCCL (ncclAllReduce through the dlopen'ed library) runs the same code."""
import threading

import numpy as np
import pytest

import scenarios

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


def run_relaxed_group(world, Xs, cfg, minibatch, tuning=None):
    from chronoclust_amd import _lib
    from chronoclust_amd.clustering.hddstream import HDDStream
    streams = [HDDStream(cfg, tuning=tuning) for _ in range(world)]
    _lib.comm_init_local([s._h for s in streams])
    for s in streams:
        s._h.comm_set_relaxed(minibatch)
    out, errors = [[] for _ in range(world)], [None] * world

    def work(rank):
        h = streams[rank]
        try:
            for t, X in enumerate(Xs):
                h.online_microcluster_maintenance(X, t)
                out[rank].append(dict(labels=h.labels_uid.copy(), paths=h.labels_path.copy(), pcore=h.table(0), outlier=h.table(1),
                                      point_cluster=h.point_cluster_index(),
                                      counters=(h.pcore_MC_last_id, h.outlier_MC_last_id),
                                      members=[c.members_in_merge_order for c in h.final_clusters],
                                      rstats=h._h.relaxed_stats()))
        except BaseException as e:  # noqa: BLE001
            errors[rank] = e
            try:
                h._h.comm_destroy()
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
    return out


def _super_steps(shard, minibatch):
    """mini-batches start at 2 048 points per rank and double up to the configured size"""
    size, pos, steps = min(minibatch, 2048), 0, 0
    while pos < shard:
        pos, steps, size = pos + size, steps + 1, min(minibatch, size * 2)
    return steps


def _same_on_all_ranks(res):
    for r in res[1:]:
        for a, b in zip(res[0], r):
            assert np.array_equal(a["labels"], b["labels"]) and np.array_equal(a["paths"], b["paths"])
            assert a["counters"] == b["counters"] and a["members"] == b["members"]
            for kind in ("pcore", "outlier"):
                for key in ("id", "uid", "w", "cf1", "cf2", "cen", "pref"):
                    assert np.array_equal(a[kind][key], b[kind][key]), (kind, key)


@pytest.mark.parametrize("world,minibatch", [(2, 1000), (4, 700), (3, 4096), (8, 512)])
def test_relaxed_group_conserves_and_agrees_with_exact(world, minibatch):
    from chronoclust_amd import multi
    from chronoclust_amd.clustering.hddstream import HDDStream
    sc = dict(seed=31, n=40_000, d=14, g=300, sigma=0.01, timepoints=3, drift=0.005, churn=0.03)
    cfg = scenarios.params_to_config(scenarios.blob_params(sc["n"], param_lambda=0.5))
    Xs = scenarios.make_blob_timepoints(sc, raw=True)
    res = run_relaxed_group(world, Xs, cfg, minibatch)
    _same_on_all_ranks(res)
    exact = HDDStream(cfg)
    for t, X in enumerate(Xs):
        exact.online_microcluster_maintenance(X, t)
        r = res[0][t]
        n = len(X)
        assert (r["labels"] >= 0).all() and not (r["paths"] & 8).any()  # every set-aside point was clustered
        uid = np.concatenate([r["pcore"]["uid"], r["outlier"]["uid"]])
        assert set(np.unique(r["labels"]).tolist()) <= set(uid.tolist())
        assert r["rstats"]["super_steps"] == _super_steps(-(-n // world), minibatch)
        assert 0 <= r["rstats"]["deferred_points"] <= n
        if t == 0:
            # no decay yet: a microcluster's weight is the number of points labelled with it, CF1 their sum
            w = np.concatenate([r["pcore"]["w"], r["outlier"]["w"]])
            u, counts = np.unique(r["labels"], return_counts=True)
            order = np.argsort(uid)
            assert np.array_equal(uid[order], u) and np.array_equal(w[order], counts.astype(np.float64))
            cf1 = np.concatenate([r["pcore"]["cf1"], r["outlier"]["cf1"]])
            for pos in np.random.default_rng(0).choice(len(uid), 20, replace=False):
                pts = X[r["labels"] == uid[pos]]
                assert np.allclose(cf1[pos], pts.sum(axis=0), rtol=1e-12, atol=0)
        # well-separated blobs: the same clusters as the exact path (points -> final cluster; outlier MCs = noise) and
        # nearly the same microclusters (a stale table lets a few points start / join other outlier MCs)
        by_cluster = multi.label_agreement(r["point_cluster"], exact.point_cluster_index())
        by_mc = multi.label_agreement(r["labels"], exact.labels_uid)
        print("world %d minibatch %d t=%d: agreement by cluster %.5f, by microcluster %.5f, set aside %d of %d" % (
            world, minibatch, t, by_cluster, by_mc, r["rstats"]["deferred_points"], n))
        assert by_cluster >= 0.995 and by_mc >= 0.97
        assert abs(len(r["members"]) - len(exact.final_clusters)) <= 2


def test_relaxed_group_of_one_rank_over_rccl():
    from chronoclust_amd import _lib, multi
    from chronoclust_amd.clustering.hddstream import HDDStream
    n, d, g = 30_000, 20, 200
    X = scenarios.make_blobs(5, n, d, g)
    cfg = scenarios.params_to_config(scenarios.blob_params(n))
    h = HDDStream(cfg)
    h._h.comm_init_rccl(_lib.comm_unique_id(), 0, 1)
    h._h.comm_set_relaxed(2048)
    h.online_microcluster_maintenance(X, 0)
    exact = HDDStream(cfg)
    exact.online_microcluster_maintenance(X, 0)
    assert (h.labels_uid >= 0).all()
    assert h._h.relaxed_stats()["super_steps"] == _super_steps(n, 2048)
    assert multi.label_agreement(h.point_cluster_index(), exact.point_cluster_index()) >= 0.995
    assert np.concatenate([h.table(0)["w"], h.table(1)["w"]]).sum() == n
    h._h.comm_set_relaxed(0)  # back to the exact path: the same handle reproduces the exact results
    h._h.reset()
    h.online_microcluster_maintenance(X, 0)
    assert np.array_equal(h.labels_uid, exact.labels_uid)


def test_c4_shaped_relaxed_two_ranks_against_exact():
    """The WNV shape of BASELINE.json (d = 14, 2 000 microclusters) at 2 M points, events sharded over two ranks in
    super-steps of up to 65 536 points per rank: ranks bit-identical, every point labelled, weights and label counts
    conserved, and - blobs being well separated - the exact path's clusters."""
    from chronoclust_amd import multi
    from chronoclust_amd.clustering.hddstream import HDDStream
    n, d, g = 2_000_000, 14, 2000
    X = scenarios.make_blobs(777, n, d, g)
    cfg = scenarios.params_to_config(scenarios.blob_params(n))
    res = run_relaxed_group(2, [X], cfg, 65536)
    _same_on_all_ranks(res)
    r = res[0][0]
    exact = HDDStream(cfg)
    exact.online_microcluster_maintenance(X, 0)
    assert (r["labels"] >= 0).all()
    uid = np.concatenate([r["pcore"]["uid"], r["outlier"]["uid"]])
    w = np.concatenate([r["pcore"]["w"], r["outlier"]["w"]])
    u, counts = np.unique(r["labels"], return_counts=True)
    order = np.argsort(uid)
    assert np.array_equal(uid[order], u) and np.array_equal(w[order], counts.astype(np.float64)) and w.sum() == n
    # a stale table lets a few early points fail a radius test the up-to-date MC would have passed: a handful of extra
    # microclusters inside their blobs, the same clusters
    assert g <= len(r["pcore"]["id"]) + len(r["outlier"]["id"]) <= 1.01 * g
    by_cluster = multi.label_agreement(r["point_cluster"], exact.point_cluster_index())
    by_mc = multi.label_agreement(r["labels"], exact.labels_uid)
    print("C4-shaped, 2 ranks: agreement by cluster %.6f, by microcluster %.6f, microclusters %d (exact %d), set aside %d" % (
        by_cluster, by_mc, len(r["pcore"]["id"]) + len(r["outlier"]["id"]), g, r["rstats"]["deferred_points"]))
    assert by_cluster >= 0.999 and by_mc >= 0.99
    assert r["rstats"]["deferred_points"] < 0.01 * n  # creation happens in the first, small super-steps only


def test_relaxed_group_with_tiny_and_uneven_timepoints():
    """Fewer points than ranks, one point, shards of unequal length, an empty last shard: every rank still takes part
    in every collective and ends with the same state; all points get labels."""
    rng = np.random.default_rng(3)
    centres = rng.uniform(0.2, 0.8, (4, 5))
    sizes = [1, 3, 1000, 7, 2]
    Xs = [np.ascontiguousarray(np.clip(centres[rng.integers(0, 4, n)] + rng.normal(0, 0.01, (n, 5)), 0, 1)) for n in sizes]
    cfg = scenarios.params_to_config(scenarios.blob_params(1000, param_lambda=0.1))
    res = run_relaxed_group(4, Xs, cfg, 256)
    _same_on_all_ranks(res)
    for t, n in enumerate(sizes):
        r = res[0][t]
        assert len(r["labels"]) == n and (r["labels"] >= 0).all()

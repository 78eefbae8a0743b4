"""Full-length oracle checks of the pruned steady state (a TOOL, run on the GPU box - minutes of host CPU; the test suite
checks prefixes only).  Test infrastructure: the oracle is the checker here, the HIP path the thing checked.

  python3 tools/full_oracle.py c2         C2 (1 M x 20, 5 000 microclusters): the WHOLE stream through the C oracle in one
                                          thread - all 10^6 labels, both tables, the merge-ordered clusters - against the HIP
                                          path's single call.
  python3 tools/full_oracle.py c5tail     a C5-shaped stream (2 M x 40, 50 000 microclusters): the last 40 % - the pruned
                                          scans on a 50 000-row table, full windows, lookahead - replayed by the oracle from
                                          the GPU's own mid-stream tables (co_inject_mc), chunk by chunk on all host cores:
                                          chunk c starts from the HIP path's state after chunk c - 1 and must reproduce its
                                          labels and its state after chunk c, bit for bit.  By induction over the chunks the
                                          chunked HIP run equals the sequential algorithm continued from the state at the
                                          half-way point; the chunked run is compared with the single call beside it.
  python3 tools/full_oracle.py c5plain    the same C5-shaped stream (2 M points) with CHRONOCLUST_HIP_PRUNE=0 and with
                                          PRUNE=2 + SCANA=2 against the default: pruned == plain, bit for bit.

Prints SHA-256 digests of what was compared, wall times and PASS / FAIL per item; exit status 1 on any FAIL.
Reference semantics: /root/reference/chronoclust/clustering/hddstream.py:220-237 (the loop), :288-462 (its body)."""
import hashlib
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenarios  # noqa: E402

KEYS = ("id", "uid", "w", "cf1", "cf2", "cen", "pref")
FAILED = []


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


def table_sha(t):
    return sha(*[t[k] for k in KEYS])


def check(name, ok, detail=""):
    print("%-4s %s%s" % ("PASS" if ok else "FAIL", name, (" - " + detail) if detail else ""), flush=True)
    if not ok:
        FAILED.append(name)


def same_tables(a, b):
    return all(np.array_equal(a[k], b[k]) for k in KEYS)


def gpu_stream(cfg, X_full):
    from chronoclust_amd.clustering.hddstream import HDDStream
    h = HDDStream(cfg)
    h._set_dataset_dependent_parameters(X_full)  # thresholds of the full timepoint (mu = mu_cfg * N)
    return h


def run_c2():
    from oracle import oracle as O
    n, d, g = 1_000_000, 20, 5000
    X = scenarios.make_blobs(42, n, d, g)
    cfg = scenarios.params_to_config(scenarios.blob_params(n))
    h = gpu_stream(cfg, X)
    t0 = time.time()
    h.online_microcluster_maintenance(X, 0, reset_param=False)
    t_gpu = time.time() - t0
    st = h.stats()
    print("C2 on the HIP path: %.1f ms online (%d windows, %d pruned scan launches of %d), wall %.2f s" % (
        st["run_ms"], st["windows"], st["scan_p_launches"], st["scan_u_launches"], t_gpu), flush=True)
    o = O.OracleHDDStream(cfg)
    o.set_dataset_dependent_parameters(X)
    t0 = time.time()
    o.online_microcluster_maintenance(X, 0, reset_param=False, offline=True)
    t_cpu = time.time() - t0
    print("C2 through the oracle, one thread: %.1f s = %.1f k points/s" % (t_cpu, n / t_cpu / 1e3), flush=True)
    check("C2 all 1 000 000 labels", np.array_equal(h.labels_uid, o.labels_uid),
          "sha256 hip %s oracle %s" % (sha(h.labels_uid), sha(o.labels_uid)))
    for kind, nm in ((0, "pcore"), (1, "outlier")):
        a, b = h.table(kind), o.table(kind)
        check("C2 %s table (%d rows: id, uid, w, cf1, cf2, cen, pref)" % (nm, len(b["id"])), same_tables(a, b),
              "sha256 hip %s oracle %s" % (table_sha(a), table_sha(b)))
    check("C2 id counters", (h.pcore_MC_last_id, h.outlier_MC_last_id) == o.counters, str(o.counters))
    got = [c.members_in_merge_order for c in h.final_clusters]
    exp = [[int(x) for x in c["members"]] for c in o.clusters]
    check("C2 clusters in merge order (%d)" % len(exp), got == exp)
    ok = all(g_.cumulative_weight == e_["w"] and np.array_equal(g_.cluster_centroids, e_["cen"])
             for g_, e_ in zip(h.final_clusters, o.clusters))
    check("C2 cluster weights and centroids", ok and len(got) == len(exp))
    print("pruned share of the HIP run: %d of %d snapshot scans pruned; start-up ends near point 200 000" % (
        st["scan_p_launches"], st["scan_u_launches"]))


def c5_stream(n):
    d, g = 40, 50_000
    X = scenarios.make_blobs(42, n, d, g)
    cfg = scenarios.params_to_config(scenarios.blob_params(n))
    return X, cfg


def run_c5tail():
    from oracle import oracle as O
    n = int(os.environ.get("N", 2_000_000))
    half = int(n * float(os.environ.get("TAIL_FROM", "0.6")))  # (40 points per blob: the table has settled by ~1.2 M, the scans are pruned from there)
    X, cfg = c5_stream(n)
    # (a one-GPU box offers 16 of its host's cores, whatever os.cpu_count() / the affinity mask report)
    workers = int(os.environ.get("WORKERS", 15))
    n_chunks = int(os.environ.get("CHUNKS", workers))
    edges = [half + (n - half) * c // n_chunks for c in range(n_chunks + 1)]
    # the single call: what a user runs
    single = gpu_stream(cfg, X)
    single.online_microcluster_maintenance(X, 0, reset_param=False)
    st = single.stats()
    print("C5-shaped (%d x 40, 50 000 blobs), single call: %.1f ms online, %d windows, %d / %d scans pruned, %d rows" % (
        n, st["run_ms"], st["windows"], st["scan_p_launches"], st["scan_u_launches"], st["rows"]), flush=True)
    # the chunked run: same stream, one call per chunk (same daystamp: no decay), state exported at every edge
    ch = gpu_stream(cfg, X)
    ch.online_microcluster_maintenance(X[:half], 0, reset_param=False)
    states = [dict(pcore=ch.table(0), outlier=ch.table(1), counters=(ch.pcore_MC_last_id, ch.outlier_MC_last_id))]
    labels = [ch.labels_uid.copy()]
    pruned = []
    for c in range(n_chunks):
        ch.online_microcluster_maintenance(X[edges[c]:edges[c + 1]], 0, reset_param=False)
        states.append(dict(pcore=ch.table(0), outlier=ch.table(1), counters=(ch.pcore_MC_last_id, ch.outlier_MC_last_id)))
        labels.append(ch.labels_uid.copy())
        s2 = ch.stats()
        pruned.append((s2["scan_p_launches"], s2["scan_u_launches"]))
    all_labels = np.concatenate(labels)
    check("chunked HIP run == single call: labels", np.array_equal(all_labels, single.labels_uid), sha(all_labels))
    for kind, nm in ((0, "pcore"), (1, "outlier")):
        check("chunked HIP run == single call: %s table" % nm, same_tables(states[-1][nm], single.table(kind)),
              table_sha(states[-1][nm]))
    print("rows at point %d: %d pcore + %d outlier; snapshot scans pruned / all, per chunk: %s" % (
        half, len(states[0]["pcore"]["id"]), len(states[0]["outlier"]["id"]), pruned), flush=True)
    # (cc_stats counts per call: every chunk's own scans - pruned ones, of all snapshot scans)
    check("the replayed stretch runs on pruned scans", all(p >= 0.9 * u and p > 0 for p, u in pruned), str(pruned[-1]))

    def replay(c):
        t0 = time.time()
        o = O.OracleHDDStream(cfg)
        o.set_dataset_dependent_parameters(X)
        o._push_params()
        s0 = states[c]
        for kind, nm in ((O.PCORE, "pcore"), (O.OUTLIER, "outlier")):
            t = s0[nm]
            for i in range(len(t["id"])):  # list order = export order
                o.inject(kind, t["cf1"][i], t["cf2"][i], t["cen"][i], t["pref"][i], t["w"][i], t["id"][i], t["uid"][i])
        inj = o.counters == s0["counters"]
        o.online_microcluster_maintenance(X[edges[c]:edges[c + 1]], 0, reset_param=False, offline=False)
        ok_l = np.array_equal(o.labels_uid, labels[c + 1])
        ok_t = same_tables(o.table(O.PCORE), states[c + 1]["pcore"]) and same_tables(o.table(O.OUTLIER), states[c + 1]["outlier"])
        ok_c = o.counters == states[c + 1]["counters"]
        return c, inj, ok_l, ok_t, ok_c, time.time() - t0, sha(o.labels_uid), table_sha(o.table(O.PCORE))

    t0 = time.time()
    from concurrent.futures import as_completed
    with ThreadPoolExecutor(workers) as ex:  # (the oracle is C behind ctypes: the GIL is released inside it)
        futs = {ex.submit(replay, c): c for c in range(n_chunks)}
        pending = set(futs)
        while pending:
            done_now = [f for f in pending if f.done()]
            if not done_now:
                time.sleep(20.0)
                print("... %d of %d chunks replayed, %.0f s" % (n_chunks - len(pending), n_chunks, time.time() - t0), flush=True)
                continue
            for f in done_now:
                pending.discard(f)
                c, inj, ok_l, ok_t, ok_c, dt, hl, ht = f.result()
                check("oracle from the HIP state at %d replays points [%d, %d): labels, both tables, counters" % (
                    edges[c], edges[c], edges[c + 1]), inj and ok_l and ok_t and ok_c,
                    "%.0f s; sha256 labels %s pcore %s" % (dt, hl, ht))
    print("oracle replay of the second half: %.0f s wall on %d threads (%d chunks)" % (time.time() - t0, workers, n_chunks))


def run_c5plain():
    from chronoclust_amd.clustering.hddstream import HDDStream
    n = int(os.environ.get("N", 2_000_000))
    X, cfg = c5_stream(n)

    def run(env):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update({k: str(v) for k, v in env.items()})
        try:
            h = HDDStream(cfg)  # (the knobs are read when the handle is created)
            t0 = time.time()
            h.online_microcluster_maintenance(X, 0)
            dt = time.time() - t0
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        st = h.stats()
        print("%-46s online %.1f ms, wall %.1f s, scans pruned %d / %d, rows %d" % (
            env or "default", st["run_ms"], dt, st["scan_p_launches"], st["scan_u_launches"], st["rows"]), flush=True)
        return h, st

    base, st0 = run({})
    check("default run prunes", st0["scan_p_launches"] > 0)
    for env in (dict(CHRONOCLUST_HIP_PRUNE=0), dict(CHRONOCLUST_HIP_PRUNE=2, CHRONOCLUST_HIP_SCANA=2),
                dict(CHRONOCLUST_HIP_PRUNE=2, CHRONOCLUST_HIP_SCANA=0)):
        h, st = run(env)
        if env.get("CHRONOCLUST_HIP_PRUNE") == 0:
            check("PRUNE=0 really ran plain scans", st["scan_p_launches"] == 0)
        ok = np.array_equal(h.labels_uid, base.labels_uid) and all(same_tables(h.table(k), base.table(k)) for k in (0, 1))
        ok = ok and [c.members_in_merge_order for c in h.final_clusters] == [c.members_in_merge_order for c in base.final_clusters]
        check("%s == default at %d x 40 / 50 000 rows: labels, tables, clusters" % (env, n), ok,
              "sha256 labels %s pcore %s" % (sha(h.labels_uid), table_sha(h.table(0))))


if __name__ == "__main__":
    what = sys.argv[1:] or ["c2"]
    for w in what:
        {"c2": run_c2, "c5tail": run_c5tail, "c5plain": run_c5plain}[w]()
    print("RESULT: %s" % ("FAIL " + "; ".join(FAILED) if FAILED else "all checks passed"))
    sys.exit(1 if FAILED else 0)

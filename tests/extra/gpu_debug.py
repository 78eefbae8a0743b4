"""Diagnostic driver for the GPU box: runs small scenarios through the HIP path and the oracle and prints the
first divergence in detail (not a test; see tests/test_hip_parity.py)."""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenarios  # noqa: E402
from chronoclust_amd.clustering.hddstream import HDDStream  # noqa: E402
from oracle import oracle as O  # noqa: E402


def compare(h, o, tag):
    ok = True
    lu, lo = h.labels_uid, o.labels_uid
    bad = np.nonzero(lu != lo)[0]
    if len(bad):
        ok = False
        i = bad[0]
        print("  [%s] labels differ at %d of %d (first %d): hip uid %d path %d | oracle uid %d path %d" % (
            tag, len(bad), len(lu), i, lu[i], h.labels_path[i], lo[i], o.paths[i]))
    pb = np.nonzero(h.labels_path != o.paths)[0]
    if len(pb):
        ok = False
        print("  [%s] paths differ at %d rows, first %d: hip %d oracle %d" % (tag, len(pb), pb[0], h.labels_path[pb[0]], o.paths[pb[0]]))
    if (h.pcore_MC_last_id, h.outlier_MC_last_id) != o.counters:
        ok = False
        print("  [%s] counters hip %s oracle %s" % (tag, (h.pcore_MC_last_id, h.outlier_MC_last_id), o.counters))
    for kind in (0, 1):
        a, b = h.table(kind), o.table(kind)
        if len(a["id"]) != len(b["id"]):
            ok = False
            print("  [%s] kind %d count hip %d oracle %d" % (tag, kind, len(a["id"]), len(b["id"])))
            continue
        for key in ("id", "uid", "w", "cf1", "cf2", "cen", "pref"):
            if not np.array_equal(a[key], b[key]):
                ok = False
                diff = np.nonzero(np.asarray(a[key] != b[key]).reshape(len(a["id"]), -1).any(axis=1))[0]
                print("  [%s] kind %d column %s differs in %d rows (first list pos %d)" % (tag, kind, key, len(diff), diff[0]))
                if key in ("w", "cf1", "cen"):
                    print("      hip", np.ravel(a[key][diff[0]])[:4], "oracle", np.ravel(b[key][diff[0]])[:4])
    got, exp = h.final_clusters, o.clusters
    if len(got) != len(exp):
        ok = False
        print("  [%s] clusters hip %d oracle %d" % (tag, len(got), len(exp)))
    else:
        for ci, (g, e) in enumerate(zip(got, exp)):
            if g.members_in_merge_order != [int(x) for x in e["members"]]:
                ok = False
                print("  [%s] cluster %d members differ" % (tag, ci), g.members_in_merge_order[:8], e["members"][:8])
                break
            if g.cumulative_weight != e["w"] or not np.array_equal(g.CF1, e["cf1"]) or not np.array_equal(g.cluster_centroids, e["cen"]) or not np.array_equal(g.preferred_dimension_vector, e["pref"]):
                ok = False
                print("  [%s] cluster %d floats differ" % (tag, ci))
                break
    return ok


def run(name, cfg, Xs, **tuning):
    print("== %s tuning=%s" % (name, tuning))
    h = HDDStream(cfg, tuning=tuning or None)
    o = O.OracleHDDStream(cfg)
    for t, X in enumerate(Xs):
        t0 = time.time()
        h.online_microcluster_maintenance(X, t)
        t1 = time.time()
        o.online_microcluster_maintenance(X, t)
        t2 = time.time()
        s = h.stats()
        ok = compare(h, o, "%s t=%d" % (name, t))
        print("  t=%d N=%d pcore=%d outlier=%d clusters=%d | hip %.3fs (run %.1f ms, windows %d rounds %d trunc %d) oracle %.3fs | %s" % (
            t, len(X), len(h.table(0)["id"]), len(h.table(1)["id"]), len(h.final_clusters), t1 - t0, s["run_ms"],
            s["windows"], s["rounds"], s["truncated"], t2 - t1, "OK" if ok else "MISMATCH"))
        if not ok:
            return False
    return True


if __name__ == "__main__":
    which = sys.argv[1:] or ["tiny", "blob", "c1"]
    try:
        if "tiny" in which:
            cfg = scenarios.params_to_config(scenarios.blob_params(500))
            run("tiny-d4-w1", cfg, [scenarios.make_blobs(1, 500, 4, 5, 0.01)], window=1)
            run("tiny-d4-w64", cfg, [scenarios.make_blobs(1, 500, 4, 5, 0.01)], window=64)
            run("tiny-d4-w1024", cfg, [scenarios.make_blobs(1, 500, 4, 5, 0.01), scenarios.make_blobs(2, 400, 4, 5, 0.01)], window=1024)
        if "blob" in which:
            cfg = scenarios.params_to_config(scenarios.blob_params(20000))
            run("blob-d20", cfg, [scenarios.make_blobs(10 + t, 20000, 20, 500, 0.01) for t in range(2)], window=1024)
        if "seed3" in which:
            n = 4000
            params = scenarios.blob_params(n, param_epsilon=0.08, param_k=4, param_pi=3)
            cfg = scenarios.params_to_config(params)
            rng = np.random.default_rng(3)
            Xs = []
            for t in range(3):
                X = scenarios.make_blobs(3 * 100 + t, n, 3, 6, 0.05)
                if t == 2:
                    X = X[rng.permutation(n)[: n // 2]]
                Xs.append(X)
            for w in (512, 64, 8):
                run("seed3-w%d" % w, cfg, Xs, window=w)
        if "c1" in which:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from golden_util import GOLDEN, StateDump
            dump = StateDump(os.path.join(GOLDEN, "c1", "hdd_state.npz"))
            extra = {k.lower(): int(v) for k, v in os.environ.items() if k in ("LOOKAHEAD", "SEGMENTS", "ROUNDS", "SEQUENTIAL")}
            run("c1", scenarios.params_to_config(scenarios.C1_PARAMS), [dump.get(t, "X") for t in range(5)],
                window=int(os.environ.get("WINDOW", "1024")), **extra)
    except Exception:
        traceback.print_exc()
        sys.exit(1)

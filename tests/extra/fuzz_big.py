"""Larger randomised parity cases than tests/test_fuzz_parity.py: tens of thousands of points, full-size windows, few to
a few hundred populations, three timepoints with drift - the regimes in which the dirty scans are not launched,
lookahead scans run and chains are long.  Everything is compared with the oracle bit for bit."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from chronoclust_amd.clustering.hddstream import HDDStream  # noqa: E402
from oracle import oracle as O  # noqa: E402


def case(seed):
    rng = np.random.default_rng(9000 + seed)
    d = int(rng.choice([2, 3, 6, 14, 20]))
    n = int(rng.choice([20000, 45000, 70000]))
    g = int(rng.choice([2, 5, 12, 40, 150, 400]))
    sigma = float(rng.choice([0.005, 0.015, 0.04]))
    cfg = {"beta": float(rng.choice([0.2, 0.5])), "delta": 0.05, "epsilon": float(rng.choice([0.04, 0.08, 0.15])),
           "lambda": float(rng.choice([0.0, 0.5, 2.0])), "k": float(rng.choice([1.0, 2.0, 3.0, 4.0])),
           "mu": float(rng.choice([10, 40])) / (0.5 * n), "pi": int(rng.choice([0, d - 1])), "omicron": float(rng.choice([0.0, 1e-4])),
           "upsilon": 6.5}
    tuning = dict(window=int(rng.choice([4096, 16384, 24576, 32768])), lookahead=int(rng.choice([0, 0, 2, 3])),
                  windows_per_sync=int(rng.choice([4, 16])))
    centres = rng.uniform(0.1, 0.9, (g, d))
    Xs = []
    for t in range(3):
        nt = n if t == 0 else int(n * rng.choice([1.0, 0.4]))
        lab = rng.integers(0, g, nt)
        Xs.append(np.ascontiguousarray(np.clip(centres[lab] + rng.normal(0, 1, (nt, d)) * sigma, 0, 1)))
        centres = np.clip(centres + rng.normal(0, 0.01, centres.shape), 0, 1)
        if g > 3:
            centres[rng.integers(0, g)] = rng.uniform(0.1, 0.9, d)  # one population moves away, a new one appears
    return cfg, tuning, Xs


if __name__ == "__main__":
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    bad = []
    for seed in range(lo, hi):
        cfg, tuning, Xs = case(seed)
        h, o = HDDStream(cfg, tuning=tuning), O.OracleHDDStream(cfg)
        ok = True
        for t, X in enumerate(Xs):
            h.online_microcluster_maintenance(X, t)
            o.online_microcluster_maintenance(X, t)
            ok = ok and np.array_equal(h.labels_uid, o.labels_uid) and (h.pcore_MC_last_id, h.outlier_MC_last_id) == o.counters
            for kind in (0, 1):
                a, b = h.table(kind), o.table(kind)
                for key in ("id", "uid", "w", "cf1", "cf2", "cen", "pref"):
                    ok = ok and np.array_equal(a[key], b[key])
            ok = ok and [c.members_in_merge_order for c in h.final_clusters] == [[int(x) for x in c["members"]] for c in o.clusters]
            if not ok:
                bad.append((seed, t, tuning))
                break
            if len(o.table(0)["id"]) + len(o.table(1)["id"]) > 6000:
                break  # (nearly every point its own microcluster: the CPU oracle needs minutes per timepoint)
        print("seed %d: %s (n %d d %d, %d MCs, windows %d)" % (seed, "ok" if ok else "MISMATCH", len(Xs[0]), Xs[0].shape[1],
                                                             len(o.table(0)["id"]) + len(o.table(1)["id"]), h.stats()["windows"]), flush=True)
    print("checked seeds %d..%d: %d bad %s" % (lo, hi - 1, len(bad), bad[:10]))

"""One-off fuzz beyond the committed seeds: labels, ids and tables against the oracle, all three lookahead modes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import test_fuzz_parity as T  # noqa: E402
from chronoclust_amd.clustering.hddstream import HDDStream  # noqa: E402
from oracle import oracle as O  # noqa: E402

if __name__ == "__main__":
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    bad = []
    for seed in range(lo, hi):
        cfg, window, Xs = T._case(seed)
        for la in (3, 0, 2):
            h = HDDStream(cfg, tuning=dict(window=window, lookahead=la, windows_per_sync=(1, 2, 5, 16)[seed % 4]))
            o = O.OracleHDDStream(cfg)
            ok = True
            for t, X in enumerate(Xs):
                h.online_microcluster_maintenance(X, t)
                o.online_microcluster_maintenance(X, t)
                ok = ok and np.array_equal(h.labels_uid, o.labels_uid) and (h.pcore_MC_last_id, h.outlier_MC_last_id) == o.counters
                for kind in (0, 1):
                    a, b = h.table(kind), o.table(kind)
                    for key in ("id", "uid", "w", "cf1", "cf2", "cen", "pref"):
                        ok = ok and np.array_equal(a[key], b[key])
                if not ok:
                    bad.append((seed, la, t, window))
                    break
    print("checked seeds %d..%d x 3 modes: %d bad %s" % (lo, hi - 1, len(bad), bad[:20]))

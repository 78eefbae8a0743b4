#!/usr/bin/env python3
"""Times the upstream Python reference on the bench generator's data (BUILD CONTAINER ONLY: it imports /root/reference
through oracle/ref_harness/refenv.py) and writes profiles/reference_py_rate.json, the `cpu_baseline.reference_py` object
of bench.py's line.  The reference cannot travel to the GPU box, so its rate is a committed measurement of this script.

    python tools/time_reference.py            # N = 10 000, d = 20, M = 100 and 400
"""
import json
import logging
import os
import platform
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle", "ref_harness"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refenv  # noqa: E402
import scenarios  # noqa: E402


def main():
    refenv.load()
    from chronoclust.clustering.hddstream import HDDStream
    n, d = 10_000, 20
    runs = []
    for g in (400, 100):
        X = scenarios.make_blobs(42, n, d, g)
        cfg = scenarios.params_to_config(scenarios.blob_params(n))
        h = HDDStream(cfg, logging.getLogger("ref"))
        t0 = time.perf_counter()
        h.online_microcluster_maintenance(X, 0)
        dt = time.perf_counter() - t0
        runs.append({"points": n, "dim": d, "microclusters": len(h.pcore_MC) + len(h.outlier_MC), "seconds": round(dt, 2),
                     "points_per_s": round(n / dt, 1)})
        print(runs[-1])
    out = {"value": [r["points_per_s"] for r in runs], "unit": "points/s", "cores": 1, "runs": runs,
           "note": "ghar1821/Chronoclust HDDStream.online_microcluster_maintenance (online + offline phase) imported "
                   "with a no-op numba stand-in, bench.py's generator at N = 10 k, d = 20, M = 400 / 100; measured by "
                   "tools/time_reference.py in the build container, not re-run on the GPU box (the reference does not "
                   "travel)",
           "host": "%s, %d usable cores, python %s, numpy %s" % (platform.processor() or platform.machine(),
                                                                len(os.sched_getaffinity(0)), platform.python_version(),
                                                                np.__version__),
           "measured_at_commit": os.popen("git -C %s rev-parse --short HEAD" % ROOT).read().strip()}
    with open(os.path.join(ROOT, "profiles", "reference_py_rate.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()

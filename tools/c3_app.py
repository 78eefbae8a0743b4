"""C3 of BASELINE.json END TO END through chronoclust_amd.app.run on one GPU: 5 timepoints x 1 M x 20 with drift, decay,
both trackers, `.npy` timepoint files in, result.csv + cluster_points_D{t}.csv + program image out.  Prints the wall
time of every phase per timepoint (chronoclust_amd.app.LAST_RUN_TIMINGS) - including write_datapoints_details, the 10^6 x 22
cell file the reference produces with DataFrame.to_csv."""
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenarios  # noqa: E402
from chronoclust_amd import app  # noqa: E402

if __name__ == "__main__":
    n = int(os.environ.get("N", 1_000_000))
    g = int(os.environ.get("G", 5000))
    normalise = os.environ.get("NORM", "0") == "1"
    sc = dict(seed=42, n=n, d=20, g=g, sigma=0.01, timepoints=5, drift=0.01, churn=0.02)
    params = scenarios.blob_params(n, param_lambda=0.5)
    tmp = tempfile.mkdtemp(dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        files = []
        for t, X in enumerate(scenarios.make_blob_timepoints(sc, raw=True)):
            fn = os.path.join(tmp, "tp%d.npy" % t)
            np.save(fn, X)
            files.append(fn)
        out = os.path.join(tmp, "out")
        os.makedirs(out)
        t0 = time.perf_counter()
        app.run(data=files, output_directory=out, normalise_data=normalise, **params)
        total = time.perf_counter() - t0
        keys = ("read", "clustering", "cluster_records", "lineage", "association", "result_rows", "point_details",
                "program_image")
        print("C3 end to end through app.run (normalise_data=%s): %.2f s for %d timepoints x %d x 20" % (
            normalise, total, len(files), n))
        for tm in app.LAST_RUN_TIMINGS:
            print("t=%d: " % tm["timepoint"] + " | ".join("%s %.0f ms" % (k, 1e3 * tm[k]) for k in keys))
        sizes = [os.path.getsize(os.path.join(out, "cluster_points_D%d.csv" % t)) for t in range(len(files))]
        print("cluster_points_D*.csv: %s MB; result.csv rows: %d" % (
            [round(s / 1e6) for s in sizes], sum(1 for _ in open(os.path.join(out, "result.csv"))) - 1))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)

"""Work per rank of the exact multi-GPU path, measured on ONE GPU: `world` handles form an in-process group (one host
thread each, the transport of tests/test_sharded_local.py) and process the same stream; all ranks share the GPU, so the wall
time of the group / world approximates what one rank's GPU has to do (replicated validation + its share of the scans +
whatever is replicated of the pruned chain) - without the links.  Compared with world = 1 it bounds the strong scaling
the 8-GPU node can show.
Usage: python tools/group_work.py [n] [d] [blobs] [worlds, e.g. 1,2,4,8]   (default: 2000000 40 50000 1,2,4,8)
Environment knobs (CHRONOCLUST_HIP_GUESS=0 ...) apply to every handle."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import numpy as np
    import pytest  # noqa: F401
    import scenarios
    import test_sharded_local as G
    import pipeline_util as P
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    g = int(sys.argv[3]) if len(sys.argv) > 3 else 50_000
    worlds = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "1,2,4,8").split(",")]
    X = scenarios.make_blobs(42, n, d, g)
    cfg = scenarios.params_to_config(scenarios.blob_params(n))
    Xs = [X, X]  # second timepoint: the steady state (table built)
    ref = None
    for world in worlds:
        t0 = time.time()
        if world == 1:
            res = [P.run_pipeline(Xs, cfg)]
        else:
            res = G.run_group(world, Xs, cfg, min_row_dims=-1, offline_min_rows=-1)
        wall = time.time() - t0
        if ref is None:
            ref = res[0]
        for r in res:
            P.same_results(r, ref)
        for t in range(len(Xs)):
            st = res[0][t]["stats"]
            on = st["run_ms"]
            print("world %d timepoint %d: online %.1f ms (%.1f ms / rank) scan %.1f ms | windows %d split %d pruned %d "
                  "guessed %d missed %d rows %d full %d" %
                  (world, t, on, on / world, st["scan_ms"], st["windows"], st["sharded_windows"], st["scan_p_launches"],
                   st["scan_g_launches"], st["missed_points"], st["pruned_scan_rows"], st["pruned_scan_full_rows"]), flush=True)
        print("world %d: wall %.2f s" % (world, wall), flush=True)


if __name__ == "__main__":
    main()

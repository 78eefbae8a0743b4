"""Records the window policy's observations and decisions (CHRONOCLUST_HIP_POLICY_TRACE) on the GPU box for the
scenarios tests/test_window_policy.py replays on the CPU; writes tests/golden/policy/<name>.jsonl (copy them back from
gpurun_out/).  Usage: python tools/record_policy_traces.py <outdir>"""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenarios  # noqa: E402
from chronoclust_amd import _lib  # noqa: E402
from chronoclust_amd.clustering.hddstream import HDDStream  # noqa: E402


def trace_to(path):
    for f in (path, path + ".rank0", path + ".rank1"):
        if os.path.exists(f):
            os.remove(f)
    os.environ["CHRONOCLUST_HIP_POLICY_TRACE"] = path


def run(cfg, Xs, tuning=None):
    h = HDDStream(cfg, tuning=tuning)
    for t, X in enumerate(Xs):
        h.online_microcluster_maintenance(X, t)
    s = h.stats()
    h._h.close()
    return s


if __name__ == "__main__":
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)
    # (a) C2: 1 M x 20, 5 000 blobs on an empty table: start-up (creation, promotion), then the steady state
    trace_to(os.path.join(out, "c2_startup_and_steady.jsonl"))
    n = 1_000_000
    print("c2", run(scenarios.params_to_config(scenarios.blob_params(n)), [scenarios.make_blobs(42, n, 20, 5000)]))
    # (b) few overlapping microclusters: windows cut short, window size oscillating, the sequential kernel taking stints
    trace_to(os.path.join(out, "few_overlapping_mcs.jsonl"))
    n = 300_000
    print("few", run(scenarios.params_to_config(scenarios.blob_params(n, param_epsilon=0.06)),
                     [scenarios.make_blobs(7, n, 5, 200, sigma=0.03)]))
    # (c) the bundled d0-d4 data: five calls, the carry from call to call, decay / downgrade between them
    from golden_util import GOLDEN, StateDump
    dump = StateDump(os.path.join(GOLDEN, "c1", "hdd_state.npz"))
    trace_to(os.path.join(out, "bundled_c1_five_timepoints.jsonl"))
    print("c1", run(scenarios.params_to_config(scenarios.C1_PARAMS), [dump.get(t, "X") for t in range(dump.n_timepoints)]))
    # (d) a group of two ranks whose table crosses the split threshold (rows x d = 400 000) while it grows: the two
    # ranks must record the same decisions - they see the same counters
    trace_to(os.path.join(out, "group_crossing_split_threshold.jsonl"))
    n, d, g = 600_000, 20, 25_000
    X = scenarios.make_blobs(7, n, d, g)
    cfg = scenarios.params_to_config(scenarios.blob_params(n))
    streams = [HDDStream(cfg) for _ in range(2)]
    _lib.comm_init_local([s._h for s in streams])
    for s in streams:
        s._h.set_shard_thresholds(400_000, 8192)
    ths = [threading.Thread(target=s.online_microcluster_maintenance, args=(X, 0)) for s in streams]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    print("group", streams[0].stats()["sharded_windows"], streams[0].stats()["windows"])
    a = open(os.path.join(out, "group_crossing_split_threshold.jsonl.rank0")).read()
    b = open(os.path.join(out, "group_crossing_split_threshold.jsonl.rank1")).read()
    assert a == b, "the ranks of a group took different decisions"
    os.rename(os.path.join(out, "group_crossing_split_threshold.jsonl.rank0"), os.path.join(out, "group_crossing_split_threshold.jsonl"))
    os.remove(os.path.join(out, "group_crossing_split_threshold.jsonl.rank1"))
    # (e) a group of two ranks at the stress config's table shape (40 dims, 50 000 microclusters: rows x d far above the
    # split threshold): split scans from early on, and pruned ones once the table has settled - the group keeps pruning
    def group_run(name, n, d, g, seed, thresholds):
        trace_to(os.path.join(out, name))
        X = scenarios.make_blobs(seed, n, d, g)
        cfg = scenarios.params_to_config(scenarios.blob_params(n))
        streams = [HDDStream(cfg) for _ in range(2)]
        _lib.comm_init_local([s._h for s in streams])
        for s in streams:
            s._h.set_shard_thresholds(*thresholds)
        ths = [threading.Thread(target=s.online_microcluster_maintenance, args=(X, 0)) for s in streams]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        st = streams[0].stats()
        print(name, "sharded", st["sharded_windows"], "of", st["windows"], "pruned launches", st["scan_p_launches"])
        a = open(os.path.join(out, name + ".rank0")).read()
        b = open(os.path.join(out, name + ".rank1")).read()
        assert a == b, "the ranks of a group took different decisions"
        os.rename(os.path.join(out, name + ".rank0"), os.path.join(out, name))
        os.remove(os.path.join(out, name + ".rank1"))
        for s in streams:
            s._h.close()

    group_run("group_c5_shape_keeps_pruning.jsonl", 1_200_000, 40, 50_000, 42, (-1, -1))
    os.environ.pop("CHRONOCLUST_HIP_POLICY_TRACE")
    for f in sorted(os.listdir(out)):
        print(f, sum(1 for _ in open(os.path.join(out, f))), "lines")

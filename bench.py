#!/usr/bin/env python3
"""bench.py — ChronoClust hot path on MI355X: points clustered per second (20-dim), config C2 of BASELINE.json.

A step = one timepoint of the hot path over one batch of synthetic input that is already resident in HBM:
reset to an empty HDDStream, the exact per-point online phase over N points (cc_online_run) and the offline
PreDeCon phase (cc_offline).  Workload at N GPUs = N independent event streams of the same shape, one per
rank ("replicas only": the online phase is a sequential chain over points and does not shard exactly, see
DESIGN.md), so scaling is weak and there is no data-path collective.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_TOPS = 39.3     # 78.6 TFLOP/s FP64 vector counts an FMA as 2; this path has no FMA


def make_blobs(seed, n, d, g, sigma=0.01):
    """BASELINE.md section 4 generator (same recipe as tests/scenarios.py)."""
    rng = np.random.default_rng(seed)
    centres = rng.uniform(0.1, 0.9, (g, d))
    lab = rng.integers(0, g, n)
    return np.ascontiguousarray(np.clip(centres[lab] + rng.normal(0.0, sigma, (n, d)), 0.0, 1.0))


def blob_config(n, beta=0.5, promote_after=10):
    return dict(beta=beta, delta=0.05, epsilon=0.05, k=4, mu=promote_after / (beta * n), pi=0, omicron=0.0,
                upsilon=6.5, **{"lambda": 0.5})


def set_params(h, cfg, n, d):
    """Derived parameters with the reference's expressions (hddstream.py:45-52, 107-126)."""
    eps = float(cfg["epsilon"])
    ups = float(cfg["upsilon"]) * eps
    delta = float(cfg["delta"])
    pi = d if float(cfg["pi"]) <= 0 else round(float(cfg["pi"]))
    h.set_params(eps ** 2, delta ** 2, float(cfg["k"]), float(cfg["beta"]), float(cfg["mu"]) * n,
                 cfg["omicron"] * 0, ups, ups ** 2, delta, pi)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=20)
    ap.add_argument("--blobs", type=int, default=5000)
    ap.add_argument("--window", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=0)
    ap.add_argument("--segments", type=int, default=0)
    ap.add_argument("--lookahead", type=int, default=0, help="0/1 on (default), 2 off, 3 forced")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=150_000)
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, one GPU per rank) or gloo (testing)")
    ap.add_argument("--share-gpu", action="store_true", help="testing: every rank uses GPU 0")
    args = ap.parse_args()

    from chronoclust_amd import multi
    rank, world, local_rank = multi.rank_info()
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.share_gpu:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.dist_backend)

    from chronoclust_amd import _lib
    n, d, g = args.points, args.dim, args.blobs
    X = make_blobs(multi.stream_seed(42, rank), n, d, g)
    cfg = blob_config(n)
    h = _lib.Handle(local_rank)
    h.set_tuning(window=args.window, rounds=args.rounds, segments=args.segments, lookahead=args.lookahead,
                 time_kernels=0 if args.no_kernel_timing else 1)
    set_params(h, cfg, n, d)
    h.points_upload(X)  # inputs are resident in HBM before the timed region

    def step():
        h.reset()
        h.online_run()
        s = h.stats()
        arrays, _ = h.offline_arrays()  # cc_offline + cc_clusters_export: every cluster's members and CF vectors on the host
        return s, len(arrays[2])

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    scan_ms = scan_launches = 0.0
    pair_dims = table_rows = 0.0
    online_ms = 0.0
    for _ in range(args.steps):
        s, n_clusters = step()
        scan_ms += s["scan_ms"]
        scan_launches += s["scan_launches"]
        pair_dims += s["scan_pair_dims"]
        table_rows += s["table_rows_scanned"]
        online_ms += s["run_ms"]
    sync()
    elapsed = time.perf_counter() - t0
    elapsed = multi.max_over_ranks(elapsed, dist, device="cuda" if args.dist_backend == "nccl" else "cpu")
    uid, _ = h.labels_download()

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    out = {
        "metric": "points clustered/sec (20-dim)" if d == 20 else "points clustered/sec (%d-dim)" % d,
        "value": multi.whole_job_rate(n, args.steps, world, elapsed),
        "unit": "points/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "C2: 1 timepoint, %dx%d synthetic blobs (seed 42+rank), %d microclusters; "
                               "exact sequential semantics; online + offline phases per step" % (n, d, g),
                   "points": n, "dim": d, "microclusters": int(s["rows"]), "clusters": n_clusters,
                   "streams": world, "window": args.window or 24576, "windows_per_step": int(s["windows"]),
                   "lookahead_windows_per_step": int(s["lookahead_windows"]),
                   "validation_rounds_per_step": int(s["rounds"]), "truncated_windows_per_step": int(s["truncated"])},
        "online_only_points_per_s": n * args.steps / (online_ms * 1e-3) if online_ms else None,
    }
    if scan_launches:
        # algorithmic traffic of one k_scan launch: its window's points (8d read + 4 label bytes per point,
        # SURVEY 8d) + the table columns the distance needs (centroid, pref: 16d + kind, key: 8 bytes per row)
        n_pts = n * args.steps
        alg_bytes = n_pts * (8 * d + 4) + table_rows * (16 * d + 8)
        secs = scan_ms * 1e-3
        achieved = alg_bytes / secs / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": "k_scan<DIRTY=false>", "achieved": achieved,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                           "launches": int(scan_launches), "avg_launch_us": 1e3 * scan_ms / scan_launches,
                           "algorithmic_bytes_per_launch": alg_bytes / scan_launches,
                           "fp64_valu": {"algorithmic_top_s": 4.0 * pair_dims / secs / 1e12,
                                         "instr_per_pair_dim": 3.0,
                                         "issued_top_s": 3.0 * pair_dims / secs / 1e12,
                                         "peak_top_s": FP64_VALU_PEAK_TOPS,
                                         "frac": 3.0 * pair_dims / secs / 1e12 / FP64_VALU_PEAK_TOPS,
                                         "note": "per (point, microcluster, dim) the reference does sub, mul, div-by-pref, "
                                                 "add (4 flops, left to right); k = 4 is a power of two, so the kernel "
                                                 "issues 3 fp64 instructions for them (v_add, v_mul, v_fma: one rounding "
                                                 "of x2 * 2^e + acc is the same double, guarded against subnormal "
                                                 "products); frac = those instructions / the 39.3 T fp64 VALU "
                                                 "instruction-lanes per second of the chip at 2.4 GHz, over ALL timed "
                                                 "launches (short start-up windows, co-running lookahead scans); the "
                                                 "11-instruction best-two update per (point, microcluster) is not "
                                                 "counted.  PMC (SQ_ACTIVE_INST_VALU / GRBM_GUI_ACTIVE) for a full "
                                                 "24 576-point launch running alone: DESIGN.md section 8"},
                           "launch_note": "every launch of the kernel is timed with HIP events on its stream: lookahead "
                                          "scans (second stream, beside the validation kernels of the previous window, "
                                          "including the few that go unused) and in-place scans (short windows of the "
                                          "start-up phase included)"}
    if "roofline" in out:
        # HBM traffic of the same kernel from the rocprofv3 PMC passes of this round (FETCH_SIZE / WRITE_SIZE in
        # separate runs, profiles/r01_pmc_traffic.json); only quoted when it was measured on this workload shape
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
                pmc = json.load(f)
            if (pmc["points"], pmc["dim"], pmc["window"]) == (n, d, out["config"]["window"]):
                out["roofline"]["traffic"] = pmc["k_scan_clean_bytes_per_launch"]
                out["roofline"]["traffic_note"] = pmc["note"]
        except (OSError, KeyError, ValueError):
            pass
    if world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        m = min(args.cpu_sample, n)
        o = O.OracleHDDStream(cfg)
        t1 = time.perf_counter()
        o.online_microcluster_maintenance(X[:m], 0, offline=False)
        # the sample runs with the full workload's thresholds (mu = mu_cfg * N), like the GPU run
        cpu_s = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": m / cpu_s, "unit": "points/s", "cores": 1, "kind": "port",
                               "sample": "first %d points of the same stream (microclusters grow 0 -> %d), "
                                         "online phase only, oracle/chrono_oracle.c single thread" % (
                                             m, o.table(O.PCORE)["id"].shape[0] + o.table(O.OUTLIER)["id"].shape[0])}
        out["cpu_baseline"]["labels_match_gpu_prefix"] = bool(np.array_equal(o.labels_uid, uid[:m]))
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Two populations whose spread sits at the preferred-dimension threshold (seed 14 of tests/test_hip_parity.py::test_long_chains_fuzz):
hundreds of heavily overlapping microclusters, every window cut short - the stream k_seq_g exists for.  Two
calls of 50 000 points; per call: wall time, windows / rounds of the windowed path, rows, points taken by the sequential kernels (of
which by k_seq_g).  Environment: D (dimensions, default 64), PI (pdim threshold, default d - 2), LA (lookahead), SEQ (cc_tuning.sequential),
CHRONOCLUST_HIP_SEQG=0 for the windowed path alone, CHRONOCLUST_HIP_LIB=<variant built with -DCC_SEQG_TIMERS -DCC_LONG_TIMERS> for the
kernel's cycles per phase."""
import sys, time, os
sys.path.insert(0,"tests"); sys.path.insert(0,".")
import numpy as np, scenarios
from chronoclust_amd.clustering.hddstream import HDDStream
seed=14
rng = np.random.default_rng(9900 + seed)
d = int(rng.choice([3, 6, 14, 20, 31, 32, 40, 64])); g = int(rng.integers(2, 60)); n = int(rng.choice([30_000, 50_000]))
sigma = float(rng.choice([0.004, 0.015, 0.03, 0.046, 0.049])); window = int(rng.choice([4096, 16384, 32768, 49152])); lookahead = int(rng.choice([0, 2, 3]))
d = int(os.environ.get("D", d)); k = float(rng.choice([1.0, 2.0, 3.0, 4.0])); eps = float(np.sqrt(float(rng.choice([1.5, 4.0])) * d * sigma * sigma / k))
cfg = scenarios.params_to_config(scenarios.blob_params(n, param_epsilon=eps, param_k=k, param_pi=(int(rng.choice([0, max(1, d - 2)])) if os.environ.get("PI") is None else int(os.environ["PI"])), param_lambda=float(rng.choice([0.0, 0.5])), promote_after=int(rng.choice([3, 10]))))
h = HDDStream(cfg, tuning=dict(window=window, lookahead=int(os.environ.get("LA", lookahead)), sequential=int(os.environ.get("SEQ","1"))))
centres = rng.uniform(0.1, 0.9, (g, d)); share = rng.dirichlet(np.full(g, 0.7))
for t in range(2):
    lab = rng.choice(g, n, p=share)
    X = np.ascontiguousarray(np.clip(centres[lab] + rng.normal(0.0, sigma, (n, d)), 0.0, 1.0))
    t0=time.time(); h.online_microcluster_maintenance(X, t); 
    s=h.stats(); print("t%d %.2f s: run_ms %.1f windows %d rounds %d truncated %d rows %d long %d prepared %d replayed %d seq %d (g %d)" % (t, time.time()-t0, s["run_ms"], s["windows"], s["rounds"], s["truncated"], s["rows"], s["long_chains"], s["long_prepared"], s["long_replayed"], s["seq_points"], s["seq_g_points"]), flush=True)
    centres = np.clip(centres + rng.normal(0.0, 0.002, centres.shape), 0.0, 1.0)

"""Scenario definitions shared by the golden generator and the parity tests.

Synthetic generator = BASELINE.md section 4 / SURVEY.md section 8(d): G blob
centres uniform in [0.1, 0.9]^d, points = centre + N(0, sigma), clipped to
[0, 1], rows already in random order.  Between timepoints every centre drifts
by N(0, drift) and a fraction of centres is retired / spawned so that the
split / new-letter / downgrade / delete paths fire.
"""
import numpy as np

# chronoclust/tests/integration_test/normal_test.py:34-51
C1_PARAMS = dict(param_beta=0.2, param_delta=0.05, param_epsilon=0.03, param_lambda=2, param_k=4, param_mu=0.01,
                 param_pi=3, param_omicron=0.000000435, param_upsilon=6.5)
# sample_run_script/sample_run.py:6-22 (BASELINE.json config 1, literally): as C1 but omicron = 4.35e-6, no gating file
SAMPLE_RUN_PARAMS = dict(param_beta=0.2, param_delta=0.05, param_epsilon=0.03, param_lambda=2, param_k=4, param_mu=0.01,
                         param_pi=3, param_omicron=0.00000435, param_upsilon=6.5)
# chronoclust/tests/integration_test/no_cluster_test.py:29-43
NOCLUSTER_PARAMS = dict(param_beta=1.0, param_delta=0.05, param_epsilon=0.03, param_lambda=2, param_k=4,
                        param_mu=1.0, param_pi=3, param_omicron=0.000000435, param_upsilon=6.5)


def params_to_config(p):
    """app.run keyword names -> the config dict HDDStream takes (app.py:95-105)."""
    return {k[len("param_"):]: v for k, v in p.items()}


def blob_params(n, beta=0.5, promote_after=10, **over):
    p = dict(param_beta=beta, param_delta=0.05, param_epsilon=0.05, param_lambda=0.5, param_k=4,
             param_mu=promote_after / (beta * n), param_pi=0, param_omicron=0.0, param_upsilon=6.5)
    p.update(over)
    return p


BLOB_SCENARIOS = {
    # d = 20, the BASELINE generator with drift; omicron > 0 so outlier deletion fires
    # (lambda = 2: retired blobs decay below beta*mu at their 2nd timepoint -> downgrade, then <= omicron -> delete)
    "d20": dict(seed=42, n=8000, d=20, g=100, sigma=0.01, timepoints=4, drift=0.01, churn=0.08,
                params=blob_params(8000, param_omicron=0.0002, param_lambda=2)),
    # d = 14 (WNV-shaped), anisotropic blobs, pi < d (pdim filter active), k not a power of two
    "d14_filter": dict(seed=7, n=5000, d=14, g=40, sigma=0.01, wide_dims=(2, 7), wide_sigma=0.08, timepoints=3,
                       drift=0.01, churn=0.05,
                       params=blob_params(5000, param_epsilon=0.25, param_pi=10, param_k=3, param_upsilon=5.6,
                                          param_omicron=0.0002, param_lambda=1.5)),
    # d = 40 (stress shape)
    "d40": dict(seed=11, n=3000, d=40, g=30, sigma=0.01, timepoints=2, drift=0.01, churn=0.1,
                params=blob_params(3000)),
    # d = 80: beyond the windowed path's 64 dimensions (k_seq_g clusters these points; d = 128 instantiations offline)
    "d80": dict(seed=13, n=2000, d=80, g=15, sigma=0.01, timepoints=3, drift=0.01, churn=0.1,
                params=blob_params(2000, param_epsilon=0.08, param_omicron=0.0002, param_lambda=1.5)),
    # normalise_data=True end to end (scaler arithmetic on the path), overlapping blobs
    "d5_norm": dict(seed=3, n=4000, d=5, g=12, sigma=0.03, timepoints=3, drift=0.02, churn=0.1, normalise=True,
                    scale=50.0,
                    params=blob_params(4000, param_epsilon=0.06, param_omicron=0.0003, param_lambda=1)),
}


def through_csv(X):
    """What the reference's pd.read_csv (default float parser, app.py:170) sees after X was written by
    DataFrame.to_csv: the parse is deterministic but not always the nearest double, so the canonical scenario
    input is the parsed array."""
    import io
    import pandas as pd
    buf = io.StringIO()
    pd.DataFrame(X, columns=["m%d" % i for i in range(X.shape[1])]).to_csv(buf, index=False)
    buf.seek(0)
    return np.ascontiguousarray(pd.read_csv(buf, header=0, sep=',').to_numpy(), dtype=np.float64)


def make_blob_timepoints(sc, raw=False):
    """raw=True: the generated arrays (what the golden generator writes to CSV);
    raw=False: those arrays after the CSV round trip (what the reference clustered)."""
    rng = np.random.default_rng(sc["seed"])
    g, d, n = sc["g"], sc["d"], sc["n"]
    centres = rng.uniform(0.1, 0.9, (g, d))
    sig = np.full((g, d), sc["sigma"])
    if sc.get("wide_dims"):
        lo, hi = sc["wide_dims"]
        for i in range(g):
            sig[i, rng.choice(d, int(rng.integers(lo, hi + 1)), replace=False)] = sc["wide_sigma"]
    out = []
    for t in range(sc["timepoints"]):
        if t > 0:
            centres = centres + rng.normal(0.0, sc["drift"], centres.shape)
            n_churn = max(1, int(round(sc["churn"] * g)))
            idx = rng.choice(g, n_churn, replace=False)
            centres[idx] = rng.uniform(0.1, 0.9, (n_churn, d))
        lab = rng.integers(0, g, n)
        X = np.clip(centres[lab] + rng.normal(0.0, 1.0, (n, d)) * sig[lab], 0.0, 1.0)
        if sc.get("scale"):
            X = X * sc["scale"]
        X = np.ascontiguousarray(X, dtype=np.float64)
        out.append(X if raw else through_csv(X))
    return out


def make_blobs(seed, n, d, g, sigma=0.01):
    """Single-timepoint BASELINE generator (bench.py uses the same recipe)."""
    rng = np.random.default_rng(seed)
    centres = rng.uniform(0.1, 0.9, (g, d))
    lab = rng.integers(0, g, n)
    return np.ascontiguousarray(np.clip(centres[lab] + rng.normal(0.0, sigma, (n, d)), 0.0, 1.0))


def blob_chunk(seed, c, rows, d, centres, sigma=0.01):
    """Rows [c * CHUNK, c * CHUNK + rows) of make_blobs_chunked(seed, ...): a chunk depends on (seed, c) only."""
    rng = np.random.default_rng([int(seed), int(c)])
    lab = rng.integers(0, centres.shape[0], rows)
    out = rng.normal(0.0, sigma, (rows, d))
    out += centres[lab]
    np.clip(out, 0.0, 1.0, out=out)
    return out


BLOB_CHUNK = 1 << 20


def blob_centres(seed, d, g):
    return np.random.default_rng(int(seed)).uniform(0.1, 0.9, (g, d))


def make_blobs_chunked(seed, n, d, g, sigma=0.01, threads=8):
    """The BASELINE generator for timepoints too large for one Generator call (C5: 50 M x 40 = 16 GB): the same
    distribution as make_blobs, generated in chunks of 2^20 rows - each from its own seeded stream, so any chunk can
    be regenerated alone - by a few threads (numpy's generators release the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    centres = blob_centres(seed, d, g)
    X = np.empty((n, d), dtype=np.float64)

    def fill(c):
        a = c * BLOB_CHUNK
        rows = min(BLOB_CHUNK, n - a)
        X[a:a + rows] = blob_chunk(seed, c, rows, d, centres, sigma)

    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(fill, range((n + BLOB_CHUNK - 1) // BLOB_CHUNK)))
    return X

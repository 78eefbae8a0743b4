"""The sequential kernels (the reference's loop taken literally: k_seq, one wavefront on an LDS image of the table; k_seq_r,
rows in registers, d <= 4; k_seq_g, one workgroup on the table in HBM once it has outgrown the image) against the goldens
of the Python reference and the oracle.  `sequential=2` forces them; a table that outgrows the LDS image in the middle
of a timepoint is handed from k_seq to k_seq_g (CHRONOCLUST_HIP_SEQG=0: back to the windowed path), so these cases also
pin the switches between the exact paths."""
import os

import numpy as np
import pytest

import scenarios
from golden_util import GOLDEN, StateDump, blob_inputs
from test_fuzz_parity import _case
from test_hip_parity import _check_against_oracle, _replay_dump

pytestmark = pytest.mark.gpu


def test_c1_golden_through_the_sequential_kernel():
    dump = StateDump(os.path.join(GOLDEN, "c1", "hdd_state.npz"))
    Xs = [dump.get(t, "X") for t in range(dump.n_timepoints)]
    h = _replay_dump(dump, Xs, scenarios.params_to_config(scenarios.C1_PARAMS), sequential=2)
    assert h.stats()["seq_points"] == len(Xs[-1])  # 143 microclusters x 3 dims fit the LDS image: the whole timepoint


def test_c1_golden_default_tuning_switches_by_itself():
    """Library defaults on the reference's own data: the windows are cut short, the sequential kernel takes over
    (and the following timepoints start on it)."""
    dump = StateDump(os.path.join(GOLDEN, "c1", "hdd_state.npz"))
    Xs = [dump.get(t, "X") for t in range(dump.n_timepoints)]
    h = _replay_dump(dump, Xs, scenarios.params_to_config(scenarios.C1_PARAMS))
    assert h.stats()["seq_points"] > 0


def test_nocluster_golden_sequential():
    dump = StateDump(os.path.join(GOLDEN, "nocluster", "hdd_state.npz"))
    Xs = [dump.get(t, "X") for t in range(dump.n_timepoints)]
    _replay_dump(dump, Xs, scenarios.params_to_config(scenarios.NOCLUSTER_PARAMS), sequential=2)


@pytest.mark.parametrize("name", sorted(scenarios.BLOB_SCENARIOS))
def test_blob_golden_sequential(name):
    """d14_filter: pdim filter + division path; d20 / d40: the table outgrows the LDS image (83 / 42 rows) after a few
    hundred points and k_seq_g finishes the timepoint on the table in HBM (with CHRONOCLUST_HIP_SEQG=0: the windowed
    path); d5_norm: stays on the image."""
    dump = StateDump(os.path.join(GOLDEN, "blob_%s.npz" % name))
    Xs = blob_inputs(name, dump)
    h = _replay_dump(dump, Xs, scenarios.params_to_config(scenarios.BLOB_SCENARIOS[name]["params"]), sequential=2)
    s = h.stats()
    assert s["seq_points"] >= 0
    if name == "d80":  # (beyond the windowed path's 64 dimensions: k_seq_g from the first point, with or without the knob)
        assert s["seq_g_points"] == len(Xs[-1])
    elif os.environ.get("CHRONOCLUST_HIP_SEQG") == "0":
        assert s["seq_g_points"] == 0
    elif name == "d20":  # (100 populations: beyond the image's 77 rows for most of the timepoint)
        assert s["seq_g_points"] > 0 and s["seq_points"] == len(Xs[-1])  # (the whole last timepoint)


@pytest.mark.parametrize("seed", range(0, 192, 2))
def test_fuzz_case_sequential(seed):
    from chronoclust_amd.clustering.hddstream import HDDStream
    from oracle import oracle as O
    cfg, window, Xs = _case(seed)
    h = HDDStream(cfg, tuning=dict(window=window, sequential=2, lookahead=3 if seed % 4 == 0 else 0))
    o = O.OracleHDDStream(cfg)
    used = 0
    for t, X in enumerate(Xs):
        h.online_microcluster_maintenance(X, t)
        o.online_microcluster_maintenance(X, t)
        _check_against_oracle(h, o)
        used += h.stats()["seq_points"]
    assert used > 0


def test_few_microclusters_long_stream_matches_oracle():
    """60 k points on 12 overlapping 5-d blobs, three timepoints with decay: sequential kernel forced, default policy
    and never - three ways through the same stream, one result."""
    from chronoclust_amd.clustering.hddstream import HDDStream
    from oracle import oracle as O
    sc = dict(seed=21, n=60_000, d=5, g=12, sigma=0.03, timepoints=3, drift=0.02, churn=0.1)
    cfg = scenarios.params_to_config(scenarios.blob_params(sc["n"], param_epsilon=0.06, param_omicron=0.0003, param_lambda=1))
    Xs = scenarios.make_blob_timepoints(sc, raw=True)
    hs = [HDDStream(cfg, tuning=dict(sequential=m)) for m in (2, 0, 1)]
    o = O.OracleHDDStream(cfg)
    for t, X in enumerate(Xs):
        o.online_microcluster_maintenance(X, t)
        for h in hs:
            h.online_microcluster_maintenance(X, t)
            _check_against_oracle(h, o)
    assert hs[0].stats()["seq_points"] > 0 and hs[2].stats()["seq_points"] == 0


@pytest.mark.parametrize("d,k,pi_off,expect_g", [(20, 4.0, 2, True), (64, 4.0, 2, True), (14, 3.0, 0, False), (40, 2.0, 0, False)])
def test_overlapping_microclusters_beyond_the_lds_image(d, k, pi_off, expect_g):
    """Two populations whose spread sits at the preferred-dimension threshold: hundreds of heavily overlapping
    microclusters, every window cut short after a handful of points - and a table beyond k_seq's LDS image.  With the
    library's defaults the policy hands the stream to k_seq_g (the table in HBM) by itself; with the pdim filter (pi < d)
    and without, k a power of two and not."""
    from chronoclust_amd.clustering.hddstream import HDDStream
    from oracle import oracle as O
    n, sigma = 12_000, 0.049
    rng = np.random.default_rng(1400 + d)
    cfg = scenarios.params_to_config(scenarios.blob_params(
        n, param_epsilon=float(np.sqrt(1.5 * d * sigma * sigma / k)), param_k=k, param_pi=(d - pi_off) if pi_off else 0,
        param_lambda=0.5, promote_after=3))
    h, o = HDDStream(cfg), O.OracleHDDStream(cfg)
    centres = rng.uniform(0.2, 0.8, (2, d))
    tot = 0
    for t in range(2):
        X = np.ascontiguousarray(np.clip(centres[rng.integers(0, 2, n)] + rng.normal(0.0, sigma, (n, d)), 0.0, 1.0))
        h.online_microcluster_maintenance(X, t)
        o.online_microcluster_maintenance(X, t)
        _check_against_oracle(h, o)
        tot += h.stats()["seq_g_points"]
    # (the first two shapes are known to end up there - tools/long_rejects.py; the others are whatever
    # the policy makes of them: parity is the test)
    if expect_g and os.environ.get("CHRONOCLUST_HIP_SEQG") != "0":
        assert tot > n, (tot, h.stats())  # (most of the stream)


# ---------------------------------------------------------------------------------------------------------
# k_seq_r (round 4): the table in registers for d <= 4 - one pcore row and a few outlier rows per lane, the 2 d
# divisions of a tentative add through one prepared reciprocal (cc_div.h)
# ---------------------------------------------------------------------------------------------------------

def _seq_r_case(seed):
    """d in 2..4, every k / pi / delta regime of the general fuzz (pdim filter, k not a power of two), populations from a
    handful to more than the kernel's 64 pcore / 256 (192 at d = 4) outlier slots hold, coarse grids (ties), coordinates
    scaled by 2^-40 or 2^-420 in some cases (the latter leaves the range k_seq_r's fast division is proven for)."""
    rng = np.random.default_rng(77_000 + seed)
    d = int(rng.choice([2, 3, 4]))
    n = int(rng.choice([40, 700, 3000, 7100]))
    g = int(rng.choice([1, 3, 8, 30, 90, 400]))
    sigma = float(rng.choice([0.0, 0.002, 0.02, 0.1]))
    grid = bool(rng.random() < 0.25)
    cfg = {
        "beta": float(rng.choice([0.1, 0.2, 0.9])),
        "delta": float(rng.choice([0.0, 0.01, 0.05, 1.0])),
        "epsilon": float(rng.choice([0.003, 0.03, 0.2])),
        "lambda": float(rng.choice([0.0, 2.0])),
        "k": float(rng.choice([0.5, 3.0, 4.0, 16.0])),
        "mu": float(rng.choice([0.0005, 0.01, 0.05])),
        "pi": int(rng.choice([1, d - 1, d])),
        "omicron": float(rng.choice([0.0, 4.35e-7, 1e-3])),
        "upsilon": 6.5,
    }
    scale = float(rng.choice([1.0, 1.0, 1.0, 2.0 ** -40, 2.0 ** -420]))  # (2^-420: below what k_seq_r's division premise admits)
    centres = rng.uniform(0.1, 0.9, (g, d))
    Xs = []
    for t in range(3):
        lab = rng.integers(0, g, n)
        X = np.clip(centres[lab] + rng.normal(0.0, 1.0, (n, d)) * sigma, 0.0, 1.0)
        if grid:
            X = np.round(X * 16) / 16
        Xs.append(np.ascontiguousarray(X * scale))
        centres = np.clip(centres + rng.normal(0, 0.01, centres.shape), 0, 1)
    if scale != 1.0:
        cfg["epsilon"] *= scale
        cfg["delta"] *= scale
    return cfg, Xs


@pytest.mark.parametrize("seed", range(96))
def test_register_resident_sequential_kernel_fuzz(seed):
    from chronoclust_amd.clustering.hddstream import HDDStream
    from oracle import oracle as O
    cfg, Xs = _seq_r_case(seed)
    h = HDDStream(cfg, tuning=dict(window=1024, sequential=2))
    o = O.OracleHDDStream(cfg)
    used = 0
    for t, X in enumerate(Xs):
        h.online_microcluster_maintenance(X, t)
        o.online_microcluster_maintenance(X, t)
        _check_against_oracle(h, o)
        used += h.stats()["seq_r_points"]
    assert used > 0


def test_register_kernel_takes_the_reference_data_and_the_knob_switches_it_off():
    """The bundled d0-d4 files (7 100 x 3): every point through k_seq_r; CHRONOCLUST_HIP_SEQR=0: through k_seq, the same
    results (the goldens of the Python reference both times)."""
    dump = StateDump(os.path.join(GOLDEN, "c1", "hdd_state.npz"))
    Xs = [dump.get(t, "X") for t in range(dump.n_timepoints)]
    cfg = scenarios.params_to_config(scenarios.C1_PARAMS)
    h = _replay_dump(dump, Xs, cfg, sequential=2)
    assert h.stats()["seq_r_points"] == h.stats()["seq_points"] == len(Xs[-1])
    old = os.environ.get("CHRONOCLUST_HIP_SEQR")
    os.environ["CHRONOCLUST_HIP_SEQR"] = "0"
    try:
        h = _replay_dump(dump, Xs, cfg, sequential=2)
    finally:
        if old is None:
            os.environ.pop("CHRONOCLUST_HIP_SEQR")
        else:
            os.environ["CHRONOCLUST_HIP_SEQR"] = old
    assert h.stats()["seq_r_points"] == 0 and h.stats()["seq_points"] == len(Xs[-1])

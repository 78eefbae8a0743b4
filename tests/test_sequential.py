"""The one-wavefront sequential kernel (k_seq: the reference's loop taken literally, table in LDS) against the
goldens of the Python reference and the oracle.  `sequential=2` forces it whenever the table fits its LDS image;
tables that outgrow the image hand the stream back to the windowed path in the middle of a timepoint, so these
cases also pin the switch between the two exact paths."""
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
    hundred points and the windowed path finishes the timepoint; d5_norm: stays sequential."""
    dump = StateDump(os.path.join(GOLDEN, "blob_%s.npz" % name))
    Xs = blob_inputs(name, dump)
    h = _replay_dump(dump, Xs, scenarios.params_to_config(scenarios.BLOB_SCENARIOS[name]["params"]), sequential=2)
    assert h.stats()["seq_points"] >= 0


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

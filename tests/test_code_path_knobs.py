"""The parity suites under every code-path knob of the library (INTEGRATION.md lists them): each knob selects kernels the
default policy would not run - or not on these inputs -, and every one of those paths has to produce the reference's
results.  Per knob: the reference-pinned golden scenarios, seeded oracle cases (with the "which kernel ran" assertions
stated per knob), a slice of the fuzz cases of tests/test_fuzz_parity.py / test_pruned_scan.py / test_sequential.py.
The knobs are read when a handle is created, so the fixture sets them around each test."""
import os

import numpy as np
import pytest

import scenarios
import test_fuzz_parity as F
import test_hip_parity as H
import test_pruned_scan as PS
import test_sequential as SQ
from golden_util import GOLDEN, StateDump, blob_inputs

pytestmark = pytest.mark.gpu

KNOBS = {
    "scan_u off": dict(CHRONOCLUST_HIP_SCANU=0),          # the LDS-staged k_scan as snapshot scan everywhere
    "long chains off": dict(CHRONOCLUST_HIP_LONGCHAINS=0),  # every chain replayed by k_chain
    "dirty scans always": dict(CHRONOCLUST_HIP_NODIRTY=0),  # the tiles' dirty scans launched in every round
    "claims by atomics": dict(CHRONOCLUST_HIP_CLAIMS=0),    # k_decide's atomics whatever the table size
    "pruning off": dict(CHRONOCLUST_HIP_PRUNE=0),
    "pruning forced": dict(CHRONOCLUST_HIP_PRUNE=2),        # k_seed / k_seed_merge / k_scan_p in every window they apply to
    "guessed thresholds off": dict(CHRONOCLUST_HIP_PRUNE=2, CHRONOCLUST_HIP_GUESS=0),
    "sparse dirty scans off": dict(CHRONOCLUST_HIP_SPARSE=0),
    "sparse dirty scans eager": dict(CHRONOCLUST_HIP_SPARSE=2),  # whenever at most every second point needs them
    "two communicators": dict(CHRONOCLUST_HIP_TWO_COMMS=1),
    "register sequential kernel off": dict(CHRONOCLUST_HIP_SEQR=0),  # k_seq (table in LDS) also for d <= 4
    "sequential kernel in HBM off": dict(CHRONOCLUST_HIP_SEQG=0),  # beyond k_seq's LDS image: the windowed path only
    "lean guessed scans off": dict(CHRONOCLUST_HIP_LEAN=0),  # k_missed and the seeded chain behind every guessed scan
    "quiet rounds off": dict(CHRONOCLUST_HIP_QUIET=0),  # k_decide re-derives every decision of every validation round
    "heavy rows off": dict(CHRONOCLUST_HIP_HEAVY=0),  # k_decide's atomics also for rows with thousands of claimants
    "creator links off": dict(CHRONOCLUST_HIP_LINK=0),  # round 0 does not link the points that decide "create": the validation rounds retarget them
    "long chains not laid out": dict(CHRONOCLUST_HIP_LONGPREP=0),  # k_chain_long alone walks them, one workgroup per chain
    # phase A of the pruned scan as a kernel of its own (k_scan_a + k_scan_p<MASKED>) whatever the table size (default: from
    # 10 000 rows on), with pruning forced so that the small tables of these suites run it at all
    "split pruned scan forced": dict(CHRONOCLUST_HIP_PRUNE=2, CHRONOCLUST_HIP_SCANA=2),
    "split pruned scan off": dict(CHRONOCLUST_HIP_PRUNE=2, CHRONOCLUST_HIP_SCANA=0),
    # round 6: the prefix test on the matrix cores (k_scan_p3) is the default; the packed-FP32 forms behind it
    "prefix test on the VALU": dict(CHRONOCLUST_HIP_PRUNE=2, CHRONOCLUST_HIP_SCANP3=0),                            # k_scan_p2
    "prefix test one point per lane": dict(CHRONOCLUST_HIP_PRUNE=2, CHRONOCLUST_HIP_SCANP3=0, CHRONOCLUST_HIP_SCANP2=0),  # k_scan_p
    "kept rows listed": dict(CHRONOCLUST_HIP_PRUNE=2, CHRONOCLUST_HIP_P3_LISTED=0),  # k_scan_p3<LISTED> whatever the table size
    "missed points through the seeded chain": dict(CHRONOCLUST_HIP_PRUNE=2, CHRONOCLUST_HIP_MISSED_PLAIN=0),
    # seeds of the seeded chain from the matrix cores + the tight threshold (k_seed16, k_seed_merge with F <= 0)
    "seeds from the matrix cores": dict(CHRONOCLUST_HIP_PRUNE=2, CHRONOCLUST_HIP_GUESS=0, CHRONOCLUST_HIP_SEED16=1),
    "no pruning outside the common case": dict(CHRONOCLUST_HIP_PRUNE=2, CHRONOCLUST_HIP_PRUNE_GENERAL=0),  # pdim filter / k != 2^e: k_scan only
}


@pytest.fixture(params=sorted(KNOBS), ids=lambda k: k.replace(" ", "_"))
def knob(request):
    kv = {k: str(v) for k, v in KNOBS[request.param].items()}
    old = {k: os.environ.get(k) for k in kv}
    os.environ.update(kv)
    yield request.param
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def test_golden_scenarios_under_knob(knob):
    """The dumps of the imported Python reference: bundled d0-d4 and the five blob scenarios (d = 80 on k_seq_g; d = 20 / 14 with the pdim
    filter / 40 / 5 normalised)."""
    dump = StateDump(os.path.join(GOLDEN, "c1", "hdd_state.npz"))
    H._replay_dump(dump, [dump.get(t, "X") for t in range(dump.n_timepoints)], scenarios.params_to_config(scenarios.C1_PARAMS),
                   window=1024)
    for name in sorted(scenarios.BLOB_SCENARIOS):
        if name == "d80" and KNOBS[knob].get("CHRONOCLUST_HIP_SEQG") == 0:
            continue  # (beyond 64 dimensions k_seq_g is the only online path: without it the call is refused)
        dump = StateDump(os.path.join(GOLDEN, "blob_%s.npz" % name))
        H._replay_dump(dump, blob_inputs(name, dump), scenarios.params_to_config(scenarios.BLOB_SCENARIOS[name]["params"]))


@pytest.mark.parametrize("case", [c for c in H.SEEDED if c[0] in (1, 3, 4, 5, 7, 10)], ids=lambda c: "seed%d" % c[0])
def test_seeded_cases_under_knob(knob, case):
    from oracle import oracle as O
    seed, n, d, g, sigma, over, window = case
    cfg = scenarios.params_to_config(scenarios.blob_params(n, **over))
    h = H._hdd(cfg, window=window, lookahead=3 if seed % 2 else 2)
    o = O.OracleHDDStream(cfg)
    tot = dict(scan_u_launches=0, scan_p_launches=0, scan_g_launches=0, windows=0)
    for t in range(3):
        X = scenarios.make_blobs(seed * 100 + t, n, d, g, sigma)
        h.online_microcluster_maintenance(X, t)
        o.online_microcluster_maintenance(X, t)
        H._check_against_oracle(h, o)
        s = h.stats()
        for k in tot:
            tot[k] += s[k]
    # which kernels ran, per knob
    k = float(cfg["k"])
    pow2 = k > 0 and np.log2(k) == round(np.log2(k))
    filter_on = 0 < float(cfg["pi"]) < d
    applies_u = pow2 and not filter_on and d in (4, 8, 14, 16, 20, 32, 40, 64)
    env = KNOBS[knob]
    # (round 6) outside the common case - the pdim filter on, k not a power of two - the pruned chain is k_scan_p3<GENERAL>
    applies_g = (not pow2 or filter_on) and d in (14, 16, 20, 32, 40) and env.get("CHRONOCLUST_HIP_SCANP3", 1) != 0 and \
        env.get("CHRONOCLUST_HIP_PRUNE_GENERAL", 1) != 0
    if tot["windows"] > 0:
        expect_u = applies_u and env.get("CHRONOCLUST_HIP_SCANU", 1) != 0
        assert (tot["scan_u_launches"] > 0) == expect_u
        if env.get("CHRONOCLUST_HIP_PRUNE") == 0 or not (expect_u or applies_g) or d <= 8:
            assert tot["scan_p_launches"] == 0
        if env.get("CHRONOCLUST_HIP_PRUNE") == 2 and (expect_u or applies_g) and d > 8:
            assert tot["scan_p_launches"] > 0
        if env.get("CHRONOCLUST_HIP_GUESS") == 0:
            assert tot["scan_g_launches"] == 0


@pytest.mark.parametrize("seed", range(0, 192, 16))
def test_fuzz_slice_under_knob(knob, seed):
    F.test_fuzz_case(seed, 3 if seed % 32 else 2)


@pytest.mark.parametrize("seed", range(0, 96, 16))
def test_pruning_fuzz_slice_under_knob(knob, seed):
    """The pruned scan's own fuzz domain (k a power of two, no filter, compiled widths > 8) - with the knob's setting of
    CHRONOCLUST_HIP_PRUNE if it has one, else forced."""
    from oracle import oracle as O
    cfg, window, lookahead, Fq, Xs = PS._fuzz_case(seed)
    prune = KNOBS[knob].get("CHRONOCLUST_HIP_PRUNE", 2)
    h = PS._hdd(cfg, prune, F=Fq, window=window, lookahead=lookahead)
    o = O.OracleHDDStream(cfg)
    for t, X in enumerate(Xs):
        h.online_microcluster_maintenance(X, t)
        o.online_microcluster_maintenance(X, t)
        PS._against_oracle(h, o)


def test_sequential_kernels_under_knob(knob):
    """The sequential kernels forced on a stream that outgrows the LDS image (k_seq, then k_seq_g - or, with that knob, the
    windows again) and chosen by the policy on heavily overlapping microclusters."""
    SQ.test_blob_golden_sequential("d20")
    SQ.test_overlapping_microclusters_beyond_the_lds_image(20, 4.0, 2, True)


def test_steady_stream_under_knob(knob):
    """A stream long enough for the policy to settle (pruned scans with guessed thresholds, lookahead, no dirty scans)
    followed by new populations: the default path's own regime changes, under the knob, against the default build's result."""
    from chronoclust_amd.clustering.hddstream import HDDStream
    n, d, g = 300_000, 20, 1500
    cfg = scenarios.params_to_config(scenarios.blob_params(n))
    rng = np.random.default_rng(5)
    X0 = scenarios.make_blobs(11, n, d, g)
    X1 = np.ascontiguousarray(np.vstack([scenarios.make_blobs(12, n // 2, d, g), scenarios.make_blobs(13, n // 2, d, 300)])[rng.permutation(n)])
    h = HDDStream(cfg)
    saved = {k: os.environ.pop(k) for k in list(KNOBS[knob]) if k in os.environ}
    try:
        ref = HDDStream(cfg)  # (the default code paths)
    finally:
        os.environ.update(saved)
    for t, X in enumerate((X0, X1)):
        h.online_microcluster_maintenance(X, t)
        ref.online_microcluster_maintenance(X, t)
        PS._same_state(h, ref)

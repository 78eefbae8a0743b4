"""CPU tests of the window policy (chronoclust_amd/csrc/cc_policy.h through cc_policy_replay): the decisions that set up
every batch of windows - window size, validation rounds, windows per batch, lookahead / dirty / pruned scans, split over
the ranks of a group - replayed from recorded device counters.

The traces under tests/golden/policy/ were recorded on an MI355X by tools/record_policy_traces.py
(CHRONOCLUST_HIP_POLICY_TRACE): every call's configuration, the counters read back after each batch, and the decision
the library took.  Replaying the counters through the policy - no GPU, no clock - must reproduce every decision: the
policy is a function of the counters alone, which is what lets the ranks of a multi-GPU group decide independently and
still enqueue the same sequence of collectives."""
import copy
import glob
import json
import os

import pytest

from chronoclust_amd import _lib

HERE = os.path.dirname(os.path.abspath(__file__))
TRACES = sorted(glob.glob(os.path.join(HERE, "golden", "policy", "*.jsonl")))
DEC_KEYS = ("win_cfg", "want", "rounds", "batch_windows", "lookahead", "nodirty", "prune", "shard", "restart", "bad", "stalled", "sparse", "probe")


def calls_of(path):
    """[(config, carry, start, first decision, [(obs, decision), ...]), ...] - one entry per online call of the trace."""
    calls = []
    for line in open(path):
        rec = json.loads(line)
        if "call" in rec:
            c = rec["call"]
            calls.append((dict(c["config"]), tuple(c["carry"]), tuple(c["start"]), c["dec"], []))
        else:
            calls[-1][4].append((rec["obs"], rec["dec"]))
    return calls


def replay(call, observations=None):
    config, carry, start, _, steps = call
    obs = [o for o, _ in steps] if observations is None else observations
    return _lib.policy_replay(config, carry, start, obs)


@pytest.mark.parametrize("path", TRACES, ids=[os.path.basename(p)[:-6] for p in TRACES])
def test_recorded_decisions_are_a_function_of_the_counters(path):
    calls = calls_of(path)
    assert calls
    carry_next = None
    for call in calls:
        config, carry, start, first, steps = call
        if carry_next is not None and config["resume"] == 0:
            assert carry == carry_next  # what the previous call left is what this one started from
        decs, carry_next = replay(call)
        assert {k: decs[0][k] for k in DEC_KEYS} == first
        for i, (_, want) in enumerate(steps):
            assert {k: decs[i + 1][k] for k in DEC_KEYS} == want, (os.path.basename(path), i)
        assert replay(call)[0] == decs  # (no hidden state)


def test_the_traces_cover_the_regimes():
    by_name = {os.path.basename(p)[:-6]: calls_of(p) for p in TRACES}
    # C2: small windows on the empty table, growth, the start-up window, then the full window, the dirty scans left out and
    # pruned snapshot scans - in place since round 6: a pruned scan on one GPU is too short for a stream of its own
    # (cc_policy.h, `short_scan`); while the scans are still plain and the stream is calm, lookahead scans run
    c2 = by_name["c2_startup_and_steady"][0]
    sizes = [c2[3]["win_cfg"]] + [d["win_cfg"] for _, d in c2[4]]
    assert sizes[0] == 256 and 4096 in sizes and sizes[-1] == 49152 and max(sizes[:4]) <= 4096  # (49 152: the default since round 6)
    last = c2[4][-1][1]
    # (prune 2: pruned scans with guessed thresholds - a mean join distance exists by then, few points are missed;
    #  prune 3: the same without the list of missed points, after a batch in which none was missed)
    assert last["lookahead"] == 0 and last["nodirty"] == 1 and last["prune"] in (2, 3)
    assert all(d["lookahead"] == 0 for _, d in c2[4] if d["prune"] != 0)
    assert any(d["prune"] == 3 for _, d in c2[4]) and any(d["prune"] == 2 for _, d in c2[4])
    assert any(d["prune"] == 0 for _, d in c2[4])  # start-up: most rows evaluated in full - the plain scan takes over
    assert any(o["stat_missed"] > 0 for o, _ in c2[4]) and all(o["tg_ok"] in (0, 1) for o, _ in c2[4])
    # few overlapping microclusters: truncated windows, the window size going down as well as up, more rounds
    few = by_name["few_overlapping_mcs"][0]
    sizes = [d["win_cfg"] for _, d in few[4]]
    assert any(b < a for a, b in zip(sizes, sizes[1:])) and any(b > a for a, b in zip(sizes, sizes[1:]))
    assert any(d["rounds"] > 1 for _, d in few[4]) and all(d["lookahead"] == 0 or d["win_cfg"] > 0 for _, d in few[4])
    # the bundled data: five calls, each starting from what the one before settled on
    assert len(by_name["bundled_c1_five_timepoints"]) == 5
    # the group: the split switches on when rows x d reaches the threshold, with a restart of the window chain
    grp = by_name["group_crossing_split_threshold"][0]
    shard = [grp[3]["shard"]] + [d["shard"] for _, d in grp[4]]
    assert shard[0] == 0 and shard[-1] == 1
    flip = shard.index(1)
    assert grp[4][flip - 1][1]["restart"] == 1 and grp[4][flip - 1][0]["m_rows"] * 20 >= 400_000
    # round 4: a group no longer trades the pruned scan for the row split - at the stress config's table shape the
    # ranks split their scans from early on and prune them once the table has settled
    c5 = by_name["group_c5_shape_keeps_pruning"][0]
    both = [d for _, d in c5[4] if d["shard"] == 1 and d["prune"] != 0]
    assert both and c5[4][-1][1]["shard"] == 1  # (24 points per microcluster: the stream ends before it has settled)
    # ... with guessed thresholds too (prune 2): the ranks agree on the missed points from the gathered records
    assert any(d["shard"] == 1 and d["prune"] in (2, 3) for _, d in c5[4])
    assert any(o["prune_rows"] > 0 for o, _ in c5[4])  # (the gathered samples of the ranks' split pruned scans)


def test_decisions_do_not_depend_on_anything_but_the_counter_deltas():
    """The same batches observed with every cumulative counter shifted by a constant (another call history) and with
    fields the policy does not read filled with noise: the same decisions."""
    call = calls_of([p for p in TRACES if "c2_startup" in p][0])[0]
    base, _ = replay(call)
    shifted = []
    for o, _ in call[4]:
        q = copy.deepcopy(o)
        q["pad"] = 12345
        shifted.append(q)
    assert replay(call, shifted)[0] == base


def test_five_batches_without_progress_stop_the_call():
    cfg = dict(window=24576, rounds_max=3, windows_per_sync=16, early_window=0, lookahead=0, allow_nodirty=1, prune_mode=2,
               prune_applicable=1, can_shard=0, d=20, resume=0, allow_sparse=1, shard_min_row_dims=400000, n_end=100000)
    stuck = dict(cursor=5000, m_rows=100, stat_windows=4, stat_tiles=16, stat_dirty_tiles=0, round_hist=[0, 4] + [0] * 8)
    obs = [dict(stuck)]
    for i in range(5):
        o = dict(stuck)
        o["stat_windows"] = 4 + 4 * (i + 1)
        obs.append(o)
    decs, _ = _lib.policy_replay(cfg, (0, 0, 1000), (0, 100), obs)
    assert [d["stalled"] for d in decs] == [0, 0, 0, 0, 0, 0, 1]
    # pruning, forced on by the caller, is given up for good at the SECOND batch in a row that commits nothing (the first one
    # may have been a batch without dirty scans: those come back first) - plain scans with the dirty scans launched always
    # decide a window's first point (round 5: seed 9348 of the forced-pruning soak, three such batches in a row)
    assert decs[1]["prune"] == 1 and decs[3]["prune"] == 0 and decs[4]["prune"] == 0 and decs[6]["prune"] == 0


def test_assumed_rate_of_the_sequential_kernel_falls_with_the_table():
    """The takeover rule of the sequential kernel compares the windows' measured rate with an assumed one until the kernel
    has been measured in the call.  For k_seq_g (tables beyond the LDS image) that rate must fall with the table size: one
    workgroup walks rows / 1 024 rows per thread, and a truncating stream on 20 000-50 000 rows whose windows still make
    30-140 points per millisecond must not be handed a 32 768-point stint of a kernel that manages a tenth of that
    (the guess used to be 150 whatever the table)."""
    g = _lib.load().cc_policy_seq_rate_guess
    # d = 20: the LDS image holds 77 rows; below that k_seq, d <= 4 the register kernel
    assert g(20, 10, 1, 1) == 700.0 and g(3, 10, 1, 1) == 1500.0 and g(3, 10, 0, 1) == 700.0
    # beyond the image: k_seq_g at its measured 8-14 us per point up to 1 024 rows ...
    assert 70.0 <= g(20, 150, 1, 1) <= 125.0 and g(20, 1024, 1, 1) == g(20, 150, 1, 1)
    # ... and in proportion to 1 024 / rows beyond
    assert g(20, 20_000, 1, 1) == pytest.approx(100.0 * 1024 / 20_000) and g(20, 50_000, 1, 1) < 30.0 / 10
    rates = [g(20, m, 1, 1) for m in (100, 1000, 2000, 5000, 20_000, 50_000)]
    assert all(a >= b for a, b in zip(rates, rates[1:]))
    # windows that still commit 30 points per millisecond on a 20 000-row table keep the stream (the rule: take over when
    # win_rate < guess); the same windows on a 300-row table hand it over
    assert not (30.0 < g(20, 20_000, 1, 1)) and 30.0 < g(20, 300, 1, 1)
    # without k_seq_g nothing takes over beyond the image; more than 64 dimensions run on it from the first row
    assert g(20, 5000, 1, 0) == 700.0 and g(100, 0, 1, 1) == 100.0
    assert g(0, 10, 1, 1) < 0 and g(20, -1, 1, 1) < 0

"""The pure-host part of the C-ABI library under the sanitizers ("race detection / sanitizers"; CPU only - GPU
AddressSanitizer is not available on this pool and is never attempted).

csrc/cc_host_abi.inc - the window policy (cc_policy_replay over csrc/cc_policy.h), the per-point text formatter
(cc_format_points_csv over csrc/cc_csv.h, called from a pool of host threads), cc_shard_rows and the sequential-kernel rate
guess: the SAME source text the product's one HIP translation unit includes - is built by g++ as a library of its own
(tests/host_san/host_abi.cpp) with -fsanitize=address,undefined and with -fsanitize=thread, and driven through the product's
own Python bindings (chronoclust_amd._lib.policy_replay / format_points_csv / shard_rows bound to the sanitizer library by
_lib.load_host_only) in a child interpreter that preloads the sanitizer runtime:

  * every recorded policy trace under tests/golden/policy/ replayed, decisions compared with the recorded ones;
  * the 65 000-value repr corpus of the formatter (specials, uniform, 60 decades, rounded, integral, subnormal) on 1 .. 16
    threads with chunk sizes down to 7 rows, compared with Python's repr;
  * cc_shard_rows over a grid of (n, world, unit) and its bad-argument paths.

A sanitizer report makes the child exit non-zero (halt_on_error / abort_on_error) and its text is the failure message."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host_san", "host_abi.cpp")

DRIVER = r'''
import glob, json, os, sys
import numpy as np
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from chronoclust_amd import _lib
_lib.load_host_only(LIB)
import test_window_policy as W

checked = 0
for path in W.TRACES:
    for call in W.calls_of(path):
        config, carry, start, first, steps = call
        decs, _ = W.replay(call)
        assert {k: decs[0][k] for k in W.DEC_KEYS} == first
        for i, (_, want) in enumerate(steps):
            assert {k: decs[i + 1][k] for k in W.DEC_KEYS} == want, (path, i)
            checked += 1
assert checked > 100, checked
# a long synthetic call: counters that wander, every branch of the policy's arithmetic under UBSan
rng = np.random.default_rng(5)
cfg = dict(window=32768, rounds_max=3, windows_per_sync=16, early_window=0, lookahead=0, allow_nodirty=1, prune_mode=1,
           prune_applicable=1, can_shard=1, d=20, resume=0, allow_sparse=128, allow_guess=1, allow_probe=1,
           shard_min_row_dims=400000, n_end=10 ** 9)
obs, cur = [], dict(cursor=0, m_rows=0, stat_windows=0, stat_truncated=0, stat_trunc_unknown=0, stat_tiles=0, stat_dirty_tiles=0,
                    stat_unsafe=0, stat_missed=0, prune_rows=0, prune_full=0, round_hist=[0] * 10)
for i in range(3000):
    wins = int(rng.integers(0, 17))
    cur = dict(cur, round_hist=list(cur["round_hist"]))
    cur["cursor"] += int(rng.integers(0, 600000)) if wins else 0
    cur["m_rows"] += int(rng.integers(0, 3000)) if rng.random() < 0.3 else 0
    cur["stat_windows"] += wins
    tr = int(rng.integers(0, wins + 1)) if rng.random() < 0.3 else 0
    cur["stat_truncated"] += tr
    cur["stat_trunc_unknown"] += int(rng.integers(0, tr + 1))
    tiles = int(rng.integers(0, 9000))
    cur["stat_tiles"] += tiles
    cur["stat_dirty_tiles"] += int(rng.integers(0, tiles + 1)) if rng.random() < 0.5 else 0
    cur["stat_unsafe"] += int(rng.integers(0, 5000)) if rng.random() < 0.5 else 0
    cur["stat_missed"] += int(rng.integers(0, 3000)) if rng.random() < 0.2 else 0
    pr = int(rng.integers(0, 10 ** 6)) if rng.random() < 0.6 else 0
    cur["prune_rows"] += pr
    cur["prune_full"] += int(rng.integers(0, pr + 1))
    cur["round_hist"][int(rng.integers(0, 4))] += wins
    o = dict(cur)
    o["stall_b"] = int(rng.integers(0, 2))
    o["tg_ok"] = int(rng.integers(0, 2))
    o["after_sequential"] = 1 if rng.random() < 0.02 else 0
    obs.append(o)
decs, carry = _lib.policy_replay(cfg, (0, 0, 1000), (0, 0), obs)
assert len(decs) == len(obs) + 1 and all(64 <= d["win_cfg"] <= 49152 for d in decs[1:])

# the formatter's corpus (tests/test_host_logic.py's, enlarged to 65 000 values) on a pool of threads
specials = [0.0, -0.0, 1.0, -1.0, 0.1, 1e-4, 9.999e-5, 1e-5, 1.5e-5, 123456789.0, 1e15, 1e16, 9999999999999998.0,
            1.2345678901234567e16, 1e22, 1e23, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, 1 / 3, 100.0,
            12345.678, 0.000123, 4.35e-07, 29.90423, 123456789012345678.0, 2.5e-310, 1e100, 1e-100, 16.0,
            float("inf"), float("-inf")]
vals = np.array(specials + list(rng.uniform(-1, 1, 20000)) + list(rng.normal(0, 1, 20000) * 10.0 ** rng.integers(-30, 30, 20000))
                + list(np.round(rng.uniform(0, 50, 15000), 5)) + list(rng.integers(-10 ** 6, 10 ** 6, 9000).astype(float))
                + list(rng.uniform(0, 1, 968) * 1e-310))
assert len(vals) == 65000
want = ["%d,L,%s" % (7 + i, repr(float(v))) for i, v in enumerate(vals)]
for threads, chunk in ((1, 65000), (2, 999), (8, 4096), (16, 7)):
    text = _lib.format_points_csv(vals.reshape(-1, 1), 7, np.zeros(len(vals), np.int32), ["L"], threads=threads, chunk=chunk).decode()
    assert text.split("\n")[:-1] == want, (threads, chunk)
Y = rng.normal(0, 1, (20000, 20))
idx = rng.integers(-1, 3, 20000).astype(np.int32)
ref = _lib.format_points_csv(Y, 0, idx, ["A|1", '"(B,C)"', "C", "None"], threads=1, chunk=20000)
import io
for threads, chunk in ((4, 512), (16, 64)):
    sink = io.BytesIO()
    assert _lib.format_points_csv(Y, 0, idx, ["A|1", '"(B,C)"', "C", "None"], threads=threads, chunk=chunk, out=sink) == len(ref)
    assert sink.getvalue() == ref
# a buffer that is too small is refused, not overrun (the caller's row bound is the formatter's own)
import ctypes as C
lib = _lib.load()
small = np.empty(64, np.uint8)
got = lib.cc_format_points_csv(Y.ctypes.data_as(C.POINTER(C.c_double)), 100, 20, 0, idx.ctypes.data_as(C.POINTER(C.c_int32)), b"ABNone",
                               np.array([0, 1, 2, 6], np.int32).ctypes.data_as(C.POINTER(C.c_int32)), 3, small.ctypes.data, small.size)
assert got == -4, got  # CC_ERR_OOM
# cc_shard_rows: blocks tile [0, n) in rank order, whole units, one block size
for n in (0, 1, 63, 64, 65, 5000, 50000, 2 ** 31 - 1):
    for world in (1, 2, 3, 8, 64):
        for unit in (1, 64):
            prev = 0
            for rank in range(world):
                lo, hi = _lib.shard_rows(n, world, rank, unit)
                assert lo == prev or (lo == n and hi == n), (n, world, unit, rank, lo, hi, prev)
                assert lo <= hi <= n and (lo % unit == 0 or lo == n)
                prev = hi
            assert prev == n
for bad in ((-1, 1, 0, 1), (10, 0, 0, 1), (10, 2, 2, 1), (10, 2, -1, 1), (10, 2, 0, 0)):
    try:
        _lib.shard_rows(*bad)
    except ValueError:
        pass
    else:
        raise AssertionError(bad)
g = lib.cc_policy_seq_rate_guess
assert g(20, 10, 1, 1) == 700.0 and g(20, 50000, 1, 1) < 3.0 and g(0, 0, 1, 1) < 0
print("HOST-SAN-OK", checked)
'''


def _runtime(name):
    out = subprocess.run(["gcc", "-print-file-name=%s" % name], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


def _build(tmp_path, flags, name):
    lib = str(tmp_path / name)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fPIC", "-shared"] + flags + ["-o", lib, SRC])
    return lib


def _drive(lib, preload, env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    env["LD_PRELOAD"] = preload
    env.pop("CHRONOCLUST_HIP_LIB", None)
    code = "ROOT = %r\nLIB = %r\n" % (ROOT, lib) + DRIVER
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "HOST-SAN-OK" in r.stdout, "exit %d\n%s\n%s" % (r.returncode, r.stdout[-3000:], r.stderr[-6000:])
    return r


def test_host_code_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("gcc has no libasan.so here")
    lib = _build(tmp_path, ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"], "libcc_host_asan.so")
    # (leaks: the interpreter's own allocations would drown the report; everything else aborts the child)
    r = _drive(lib, asan, dict(ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",
                               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1"))
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]


def test_threaded_formatter_under_thread_sanitizer(tmp_path):
    tsan = _runtime("libtsan.so")
    if tsan is None:
        pytest.skip("gcc has no libtsan.so here")
    lib = _build(tmp_path, ["-fsanitize=thread"], "libcc_host_tsan.so")
    probe = subprocess.run([sys.executable, "-c", "print('ok')"], capture_output=True, text=True,
                           env=dict(os.environ, LD_PRELOAD=tsan, TSAN_OPTIONS="report_bugs=0"))
    if probe.returncode != 0 or "ok" not in probe.stdout:
        pytest.skip("this interpreter does not start under the ThreadSanitizer runtime: %s" % probe.stderr[-300:])
    r = _drive(lib, tsan, dict(TSAN_OPTIONS="halt_on_error=1:exitcode=66:second_deadlock_stack=1"))
    assert "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-4000:]

"""The ONE stdout line of bench.py (the driver parses it; round 4's 29 KB line could not be parsed): built from a canned
full record - round 4's own line, profiles/r04_bench.json -, it stays below bench.LINE_LIMIT bytes whatever the legs put
into their objects, round-trips through json, carries the contract's keys with `roofline` and `cpu_baseline`, and is
the last (and only) line the keeper process writes - also into a non-blocking pipe."""
import copy
import fcntl
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def canned():
    with open(os.path.join(ROOT, "profiles", "r04_bench.json")) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_round4_record_becomes_a_compact_line():
    full = canned()
    assert len(json.dumps(full)) > 20000  # (the record that was too long for the driver)
    text = bench.compact_line(full, "bench_detail.json")
    assert len(text) < bench.LINE_LIMIT and "\n" not in text
    line = json.loads(text)
    for k in CONTRACT:
        assert k in line, k
    assert line["value"] == pytest.approx(full["value"], rel=1e-5)
    assert line["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    assert set(line["config"]) >= {"workload", "points", "dim", "microclusters"}
    rf = line["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "valu_busy", "effective_frac", "hbm_frac", "traffic",
              "algorithmic_bytes", "launches", "avg_launch_us", "pmc_source"):
        assert k in rf, k
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-4)
    # (since round 6 the plain scan of the start-up windows is the scan kernel that takes most of a step's scan time: its figure)
    assert rf["valu_busy"] == pytest.approx(full["roofline"]["executed"]["kernels"]["k_scan_u"]["valu_busy_fraction"], rel=1e-5)
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["sample"]
    for name in bench.LEG_NAMES:
        leg = line["strong_scaling"][name]
        assert leg["n_gpus"] == 1 and leg["rccl_ranks_seen"] == [1] and leg["all_ranks_bit_identical"] is True
        assert name not in line  # (the legs' own objects are in the detail file)
    assert line["legs_failed"] == [] and line["detail_file"] == "bench_detail.json"


def test_line_stays_small_whatever_the_legs_say():
    full = canned()
    full["config"]["workload"] = "w" * 5000
    full["roofline"]["kernel"] = "k" * 5000
    full["roofline"]["unit"] = "u" * 5000
    full["cpu_baseline"]["sample"] = "s" * 5000
    for name in bench.LEG_NAMES:
        full[name] = {"error": "RuntimeError: " + "x" * 5000}
        full["strong_scaling"][name]["rccl_ranks_seen"] = list(range(8))
        full["strong_scaling"][name]["n_gpus"] = 8
    full["legs_failed"] = list(bench.LEG_NAMES)
    full["incomplete"] = "i" * 5000
    text = bench.compact_line(full, "/some/long/path/" + "d" * 200 + "/bench_detail.json")
    assert len(text) < bench.LINE_LIMIT
    line = json.loads(text)
    assert line["legs_failed"] == list(bench.LEG_NAMES) and set(line["leg_errors"]) == set(bench.LEG_NAMES)
    for k in CONTRACT:
        assert k in line, k


def test_headline_without_legs_or_counters():
    """What rank 0 holds right after the headline (no leg yet, PMC files of other kernel sources): still a valid line."""
    full = copy.deepcopy(canned())
    for name in bench.LEG_NAMES:
        full.pop(name, None)
        full["strong_scaling"].pop(name, None)
    full["roofline"]["executed"] = None
    full["roofline"]["achieved"] = full["roofline"]["frac"] = None
    line = json.loads(bench.compact_line(full, None))
    assert line["roofline"]["frac"] is None and line["roofline"]["effective_frac"] > 0 and line["detail_file"] is None


KEEPER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, %r)
    import bench
    full = json.loads(open(%r).read().strip().splitlines()[-1])
    os.environ["X"] = "1"
    bench.DETAIL_FILE = sys.argv[1]
    sys.stdout.flush()
    fd = os.dup(1)
    os.dup2(2, 1)
    g = bench.LineGuard(fd)
    g.provisional(full)
    print("noise on fd 1 goes to stderr")
    if sys.argv[2] == "final":
        full["legs_failed"] = ["x"]
        g.final(full)
    else:
        os._exit(7)   # the process dies before the final line: the keeper prints the provisional one
""") % (ROOT, os.path.join(ROOT, "profiles", "r04_bench.json"))


@pytest.mark.parametrize("mode", ["final", "dies"])
@pytest.mark.parametrize("nonblocking", [False, True])
def test_keeper_prints_exactly_one_line(tmp_path, mode, nonblocking):
    detail = str(tmp_path / "detail.json")
    r, w = os.pipe()
    if nonblocking:
        fcntl.fcntl(w, fcntl.F_SETFL, fcntl.fcntl(w, fcntl.F_GETFL) | os.O_NONBLOCK)
    p = subprocess.Popen([sys.executable, "-c", KEEPER, detail, mode], stdout=w, stderr=subprocess.DEVNULL)
    os.close(w)
    data = b""
    while True:
        chunk = os.read(r, 65536)
        if not chunk:
            break
        data += chunk
    p.wait()
    lines = data.decode().strip().splitlines()
    assert len(lines) == 1 and len(lines[0]) < bench.LINE_LIMIT
    line = json.loads(lines[0])
    assert line["detail_file"] == detail
    if mode == "final":
        assert p.returncode == 0 and line["legs_failed"] == ["x"] and "incomplete" not in line
    else:
        assert p.returncode == 7 and "incomplete" in line
    with open(detail) as f:
        full = json.load(f)
    assert "executed" in full["roofline"] and "one_stream_exact" in full  # (the detail keeps everything)

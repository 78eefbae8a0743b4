"""VALU / LDS utilisation of the snapshot-scan kernels from three rocprofv3 --pmc passes over tools/steady.py (LA=2: scans run alone).
Usage: pmc_valu_summary.py <dir of pass a: GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU>
                           <dir of pass b: GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES>
                           <dir of pass c: GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT, or ->
                           <d> <rows> [window] [stdout of a steady.py run: its share of rows evaluated in full]
The output carries the SHA-256 of chronoclust_amd/csrc/: bench.py quotes these figures (roofline.executed) only for the
kernel sources they were measured on."""
import glob
import json
import os
import sys

import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (csrc_digest)

D, ROWS = int(sys.argv[4]), int(sys.argv[5])
WINDOW = int(sys.argv[6]) if len(sys.argv) > 6 else 32768  # the library's default window


def full_launches(d, pat):
    if d == "-":
        return None, 0
    f = (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0]
    df = pd.read_csv(f)
    df = df[df["Kernel_Name"].str.contains(pat, regex=False)]
    if not len(df):
        return None, 0
    g = df.groupby(["Dispatch_Id", "Counter_Name"])["Counter_Value"].sum().unstack()
    t = df.groupby("Dispatch_Id").agg(s=("Start_Timestamp", "first"), e=("End_Timestamp", "first"), grid=("Grid_Size", "first"))
    g["us"] = (t["e"] - t["s"]) / 1e3
    g["grid"] = t["grid"]
    # the full-window launches: the largest grid that at least a tenth of the dispatches have (k_scan_p also runs on small
    # grids - the points a guessed threshold missed, probes -, k_seed / k_seed_merge in the steady state ONLY on those:
    # their figures are then what a window really spends on them), and of those the typical ones - within a factor of
    # 1.5 of the median duration (a launch whose window was scanned ahead returns at once; the launches of the build-up
    # run, where every other row still survives its prefix, take several times as long as the steady state's)
    counts = g["grid"].value_counts()
    big = max(k for k, v in counts.items() if v >= max(1, len(g) // 10))
    common = g[g["grid"] == big]
    med = common["us"].median()
    full = common[(common["us"] > med / 1.5) & (common["us"] < med * 1.5)]
    return full.mean(), int(len(full))


FULL = None
if len(sys.argv) > 7 and os.path.exists(sys.argv[7]):
    import re
    hits = re.findall(r"([0-9.]+) % evaluated in full", open(sys.argv[7]).read())
    if hits:
        FULL = float(hits[-1]) / 100.0

out = {"shape": "%d points x %d microclusters x %d dims per launch, running alone (tools/steady.py, LA=2)" % (WINDOW, ROWS, D),
       "csrc_sha256": bench.csrc_digest(), "scan_sha256": bench.scan_digest(), "rows_evaluated_in_full_frac": FULL, "dim": D, "rows": ROWS, "window": WINDOW, "kernels": {}}
# (k_scan_p3 / k_prefix16: rocprofv3 leaves these instances mangled - matched without the "<")
for pat in ("k_seed<", "k_seed_merge", "k_scan_a<", "k_scan_p<", "k_scan_p2<", "k_prefix16", "k_scan_p3", "k_scan_u<"):
    a, na = full_launches(sys.argv[1], pat)
    b, nb = full_launches(sys.argv[2], pat)
    c, nc = full_launches(sys.argv[3], pat)
    if a is None or b is None or na == 0 or nb == 0:
        continue
    cycles = a["GRBM_GUI_ACTIVE"] / 8.0  # the counter sums the 8 XCDs
    pairs = WINDOW / 64 * ROWS           # (wave of 64 points, table row) pairs a launch covers
    k = {
        "launches": [na, nb, nc],
        "avg_us_under_pmc": [float(a["us"]), float(b["us"])] + ([float(c["us"])] if c is not None else []),
        # (a launch that ends within a few tens of us is mostly dispatch: GRBM cycles / its duration is not a clock)
        "effective_clock_ghz": float(cycles / (a["us"] * 1e3)) if a["us"] >= 50.0 else None,
        "valu_instructions_per_wave_row": float(a["SQ_INSTS_VALU"] / pairs),
        "salu_instructions_per_wave_row": float(a["SQ_INSTS_SALU"] / pairs),
        "valu_busy_fraction": float(a["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / cycles),
        "wave_time_issuing": float(b["SQ_ACTIVE_INST_ANY"] / b["SQ_WAVE_CYCLES"]),
        "wave_time_waiting_to_issue": float(b["SQ_WAIT_INST_ANY"] / b["SQ_WAVE_CYCLES"]),
        "wave_time_parked_on_waitcnt": float(b["SQ_WAIT_ANY"] / b["SQ_WAVE_CYCLES"]),
    }
    if c is not None and nc > 0:
        ccyc = c["GRBM_GUI_ACTIVE"] / 8.0
        k["lds_instructions_per_wave_row"] = float(c["SQ_INSTS_LDS"] / pairs)
        k["lds_busy_fraction"] = float(c["SQ_ACTIVE_INST_LDS"] * 4.0 / 1024.0 / ccyc)   # SIMD-cycles with an LDS instruction active
        k["lds_array_busy_fraction"] = float(c["SQ_LDS_IDX_ACTIVE"] / 256.0 / ccyc)  # LDS-array cycles per CU and cycle
        k["lds_bank_conflict_fraction"] = float(c["SQ_LDS_BANK_CONFLICT"] / max(1.0, c["SQ_LDS_IDX_ACTIVE"]))
    # what keeps the kernel from the VALU issue roof, named from the counters
    parked, lds = k["wave_time_parked_on_waitcnt"], k.get("lds_busy_fraction", 0.0)
    if k["valu_busy_fraction"] >= 0.8:
        k["co_limiter"] = "none: VALU issue"
    elif k.get("lds_array_busy_fraction", 0.0) >= 0.3:
        k["co_limiter"] = "LDS data return (wave-uniform row reads) beside the VALU"
    elif parked >= 0.5:
        k["co_limiter"] = "memory latency (waves parked on s_waitcnt)"
    else:
        k["co_limiter"] = "scalar-load waits and wave-launch overheads"
    out["kernels"][pat.rstrip("<")] = k
if ("k_scan_p" in out["kernels"] or "k_scan_p2" in out["kernels"] or "k_scan_p3" in out["kernels"]) and "k_scan_u" in out["kernels"]:
    # (with the pruned scan on, k_scan_u only ran on the short windows of the build-up run: its per-row figures would
    # be scaled by the wrong window; the plain scan's own are in the *_plain.json file, measured with CHRONOCLUST_HIP_PRUNE=0)
    del out["kernels"]["k_scan_u"]
out["note"] = ("three PMC passes; SQ_* count quad-cycles; valu_busy_fraction = SQ_ACTIVE_INST_VALU x 4 cycles / 1024 "
               "SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs), lds_busy_fraction likewise from SQ_ACTIVE_INST_LDS, lds_array_busy_fraction = "
               "SQ_LDS_IDX_ACTIVE / 256 CUs / cycles (LDS-array cycles: a wave-uniform ds_read_b128 takes 4); instructions per (wave, row): the plain scan k_scan_u spends 3 d of "
               "them on the distance terms alone (60 at d = 20, 120 at d = 40)")
print(json.dumps(out, indent=1))

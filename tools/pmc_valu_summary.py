"""VALU utilisation of the snapshot-scan kernels from two rocprofv3 --pmc passes over tools/steady.py (LA=2: scans run alone).
Usage: pmc_valu_summary.py <dir of pass a: GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU>
                           <dir of pass b: GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES> <d> <rows> [window]"""
import glob
import json
import sys

import pandas as pd

D, ROWS = int(sys.argv[3]), int(sys.argv[4])
WINDOW = int(sys.argv[5]) if len(sys.argv) > 5 else 32768  # the library's default window


def full_launches(d, pat):
    f = (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0]
    df = pd.read_csv(f)
    df = df[df["Kernel_Name"].str.contains(pat, regex=False)]
    if not len(df):
        return None, 0
    g = df.groupby(["Dispatch_Id", "Counter_Name"])["Counter_Value"].sum().unstack()
    t = df.groupby("Dispatch_Id").agg(s=("Start_Timestamp", "first"), e=("End_Timestamp", "first"), grid=("Grid_Size", "first"))
    g["us"] = (t["e"] - t["s"]) / 1e3
    g["grid"] = t["grid"]
    # the full-window launches: the most frequent grid, and of those the typical ones - within a factor of 1.5 of the
    # median duration (a launch whose window was scanned ahead returns at once; the launches of the build-up run, where
    # every other row still survives its prefix, take several times as long as the steady state's)
    common = g[g["grid"] == g["grid"].mode().iloc[0]]
    med = common["us"].median()
    full = common[(common["us"] > med / 1.5) & (common["us"] < med * 1.5)]
    return full.mean(), int(len(full))


out = {"shape": "%d points x %d microclusters x %d dims per launch, running alone (tools/steady.py, LA=2)" % (WINDOW, ROWS, D),
       "kernels": {}}
for pat in ("k_seed<", "k_seed_merge", "k_scan_p<", "k_scan_u<"):
    a, na = full_launches(sys.argv[1], pat)
    b, nb = full_launches(sys.argv[2], pat)
    if a is None or b is None or na == 0 or nb == 0:
        continue
    cycles = a["GRBM_GUI_ACTIVE"] / 8.0  # the counter sums the 8 XCDs
    pairs = WINDOW / 64 * ROWS           # (wave of 64 points, table row) pairs a launch covers
    out["kernels"][pat.rstrip("<")] = {
        "launches": [na, nb],
        "avg_us_under_pmc": [float(a["us"]), float(b["us"])],
        "effective_clock_ghz": float(cycles / (a["us"] * 1e3)),
        "valu_instructions_per_wave_row": float(a["SQ_INSTS_VALU"] / pairs),
        "salu_instructions_per_wave_row": float(a["SQ_INSTS_SALU"] / pairs),
        "valu_busy_fraction": float(a["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / cycles),
        "wave_time_issuing": float(b["SQ_ACTIVE_INST_ANY"] / b["SQ_WAVE_CYCLES"]),
        "wave_time_waiting_to_issue": float(b["SQ_WAIT_INST_ANY"] / b["SQ_WAVE_CYCLES"]),
        "wave_time_parked_on_waitcnt": float(b["SQ_WAIT_ANY"] / b["SQ_WAVE_CYCLES"]),
    }
if "k_scan_p" in out["kernels"] and "k_scan_u" in out["kernels"]:
    # (with the pruned scan on, k_scan_u only ran on the short windows of the build-up run: its per-row figures would
    # be scaled by the wrong window; the plain scan's own are in the *_plain.json file, measured with CHRONOCLUST_HIP_PRUNE=0)
    del out["kernels"]["k_scan_u"]
out["note"] = ("two PMC passes of five counters; SQ_* count quad-cycles; valu_busy_fraction = SQ_ACTIVE_INST_VALU x 4 cycles / 1024 "
               "SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs); instructions per (wave, row): the plain scan k_scan_u spends 3 d of them on "
               "the distance terms alone (60 at d = 20, 120 at d = 40)")
print(json.dumps(out, indent=1))

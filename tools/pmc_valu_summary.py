"""VALU utilisation of the snapshot scan (k_scan_u) from two rocprofv3 --pmc passes over tools/steady.py (LA=2: scans run alone).
Usage: pmc_valu_summary.py <dir of pass a: GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU>
                           <dir of pass b: GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES>"""
import glob
import json
import sys

import pandas as pd


def full_launches(d):
    f = (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0]
    df = pd.read_csv(f)
    df = df[df["Kernel_Name"].str.contains("k_scan_u<20, 4>", regex=False)]
    g = df.groupby(["Dispatch_Id", "Counter_Name"])["Counter_Value"].sum().unstack()
    t = df.groupby("Dispatch_Id").agg(s=("Start_Timestamp", "first"), e=("End_Timestamp", "first"), grid=("Grid_Size", "first"))
    g["us"] = (t["e"] - t["s"]) / 1e3
    g["grid"] = t["grid"]
    full = g[(g["grid"] == g["grid"].max()) & (g["us"] > 0.6 * g["us"].max())]
    return full.mean(), int(len(full))


a, na = full_launches(sys.argv[1])
b, nb = full_launches(sys.argv[2])
cycles = a["GRBM_GUI_ACTIVE"] / 8.0  # the counter sums the 8 XCDs
out = {
    "kernel": "k_scan_u<20, 4>, 24576 points x 5000 microclusters, running alone",
    "launches": [na, nb],
    "avg_us_under_pmc": [float(a["us"]), float(b["us"])],
    "effective_clock_ghz": float(cycles / (a["us"] * 1e3)),
    "valu_instructions_per_wave_row": float(a["SQ_INSTS_VALU"] / (24576 / 64 * 5000)),
    "valu_busy_fraction": float(a["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / cycles),
    "wave_time_issuing": float(b["SQ_ACTIVE_INST_ANY"] / b["SQ_WAVE_CYCLES"]),
    "wave_time_waiting_to_issue": float(b["SQ_WAIT_INST_ANY"] / b["SQ_WAVE_CYCLES"]),
    "wave_time_parked_on_waitcnt": float(b["SQ_WAIT_ANY"] / b["SQ_WAVE_CYCLES"]),
    "note": "two PMC passes of at most five counters; SQ_* count quad-cycles; valu_busy_fraction = SQ_ACTIVE_INST_VALU x 4 "
            "cycles / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)",
}
print(json.dumps(out, indent=1))

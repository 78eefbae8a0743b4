"""HBM-side traffic of the snapshot scan (k_scan_u) from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs of
`bench.py --steps 1 --warmup 0 --no-cpu-baseline`).  Usage: pmc_summary.py <fetch_dir> <write_dir> <window> > json"""
import glob
import json
import sys

import pandas as pd


def per_launch(d, counter):
    f = (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0]
    df = pd.read_csv(f)
    df = df[df["Counter_Name"] == counter]
    df = df[df["Kernel_Name"].str.contains("k_scan_u<20, 4>", regex=False)]
    g = df.groupby("Dispatch_Id")["Counter_Value"].sum()
    name = df["Kernel_Name"].iloc[0].split("(")[0]
    # every launch of the kernel, like bench.py's avg_launch_us and algorithmic_bytes_per_launch (lookahead batches
    # also enqueue the in-place scan, which returns at once when the window was scanned ahead)
    return float(g.mean()), int(len(g)), name


if __name__ == "__main__":
    fetch_kb, n1, name = per_launch(sys.argv[1], "FETCH_SIZE")
    write_kb, n2, _ = per_launch(sys.argv[2], "WRITE_SIZE")
    out = {"points": 1000000, "dim": 20, "window": int(sys.argv[3]), "kernel": name, "launches": n1,
           "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
           "k_scan_clean_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
           "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of `bench.py --steps 1 --warmup 0`; "
                   "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (MI355X_MICROARCH.md: gfx950 FETCH_SIZE reports half of "
                   "a wide coalesced read; narrower accesses are uncalibrated, so this is an upper estimate); includes "
                   "Infinity-Cache hits; averaged over all launches of the kernel in the run (the start-up phase uses short "
                   "windows, a few lookahead scans go unused). Most of it is the per-workgroup argmin partials (window x 8..16 x 64 B) that the "
                   "scan writes and k_decide merges, not input data; scalar-cache fills of the row operands are not in these counters."}
    print(json.dumps(out, indent=1))
